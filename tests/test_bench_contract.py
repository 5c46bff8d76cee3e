"""The benchmark's command line and its one-line JSON result, as the round driver reads them."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run_bench(*args, timeout=300):
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *args], cwd=ROOT, env=env,
                       capture_output=True, text=True, timeout=timeout)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]
    # ... and nothing else on stdout (native libraries' banners go to stderr)
    assert [ln for ln in p.stdout.splitlines() if ln.strip()] == lines, p.stdout[:2000]
    return json.loads(lines[0])


def test_traffic_of_another_configuration_is_null_not_borrowed(monkeypatch):
    """A PMC figure is keyed by workload, batch and point distribution: anything else reads as None."""
    sys.path.insert(0, ROOT)
    import bench
    from tools import buildinfo
    t = json.load(open(os.path.join(ROOT, "profiles", "traffic.json")))
    # (as if the sources were still the ones the record was measured on)
    monkeypatch.setattr(buildinfo, "csrc_hash", lambda: t["T-fwdbwd|batch=1|points=uniform"]["_csrc_sha256"])
    assert bench.load_traffic("T-fwdbwd", 1, "uniform", "msmv_bwd_owner_dfeat")["bytes"] > 0
    assert bench.load_traffic("T-fwdbwd", 8, "uniform", "msmv_bwd_owner_dfeat") is None
    assert bench.load_traffic("T-fwdbwd", 1, "cameras", "msmv_bwd_owner_dfeat") is None
    assert bench.load_traffic("T-fwdbwd", 1, "uniform", "no_such_kernel") is None


def test_traffic_measured_on_other_kernel_sources_is_null(monkeypatch):
    """Every record carries the hash of the kernel sources it was measured on; once they change the
    stale figure is no longer reported."""
    sys.path.insert(0, ROOT)
    import bench
    from tools import buildinfo
    t = json.load(open(os.path.join(ROOT, "profiles", "traffic.json")))
    measured = [k for k, v in t.items() if isinstance(v, dict)]
    assert measured and all("_csrc_sha256" in t[k] for k in measured)
    monkeypatch.setattr(buildinfo, "csrc_hash", lambda: "0" * 16)
    assert bench.load_traffic("T-fwdbwd", 1, "uniform", "msmv_bwd_owner_dfeat") is None


def test_help_lists_the_contract_flags():
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--help"], cwd=ROOT,
                       capture_output=True, text=True, timeout=120)
    assert p.returncode == 0
    for flag in ("--gpus", "--steps", "--warmup", "--workload"):
        assert flag in p.stdout


@pytest.mark.gpu
def test_result_line_carries_every_contract_field():
    r = run_bench("--gpus", "1", "--steps", "4", "--warmup", "2", "--workload", "T-fwdbwd")
    assert r["n_gpus"] == 1 and r["steps"] == 4 and r["warmup"] == 2
    assert r["unit"] == "sampled points/s" and r["higher_is_better"] is True
    assert r["scaling"] == "weak" and r["vs_baseline"] is None
    assert r["dtype"] == "f32" and r["data"] == "synthetic"
    assert "OPUS-T" in r["config"]["workload"] and "model" not in r["config"]
    assert r["value"] > 0 and r["ms_per_step"] > 0
    # value = points of all ranks / elapsed; one rank, 32 slabs x 600 queries x 4 points per step
    assert abs(r["value"] - 32 * 600 * 4 / (r["ms_per_step"] * 1e-3)) <= 1e-6 * r["value"]
    roof = r["roofline"]
    assert roof["bound"] == "hbm" and roof["unit"] == "GB/s" and roof["peak"] == 8000.0
    # achieved counts compulsory bytes: a true fraction of the peak; the no-reuse figure sits beside it
    assert roof["achieved"] > 0 and abs(roof["frac"] - roof["achieved"] / roof["peak"]) < 1e-9
    assert roof["frac"] <= 1.0 and roof["achieved_alg"] >= roof["achieved"]
    k = r["kernels"][roof["kernel"]]
    # ... over the MEDIAN of at least 20 evented launches, whatever --steps is
    assert k["launches"] >= 20 and roof["launches"] == k["launches"]
    assert abs(roof["achieved"] - k["compulsory_bytes"] / (k["p50_us"] * 1e-6) / 1e9) <= 1e-6 * roof["achieved"]
    worst = r["roofline_worst"]
    assert worst["frac"] <= roof["frac"] + 1e-12 and worst["share_of_kernel_time"] >= 0.1
    # traffic: the committed PMC pass of exactly this workload | batch | points (profiles/traffic.json),
    # reported only while the kernel sources are the ones it was measured on
    sys.path.insert(0, ROOT)
    from tools.buildinfo import csrc_hash
    with open(os.path.join(ROOT, "profiles", "traffic.json")) as f:
        conf = json.load(f)["T-fwdbwd|batch=1|points=uniform"]
    rec = conf[roof["kernel"]]
    if conf["_csrc_sha256"] == csrc_hash():
        assert roof["traffic"] == rec["bytes"] and rec["template"].startswith(roof["kernel"])
        assert roof["traffic"] <= 1.1 * k["compulsory_bytes"], "the dominant kernel moves what it must, not more"
    else:
        assert roof["traffic"] is None
    # the whole-pass algorithmic bytes are the contract's: 4380 B/point forward, 8504 B/point + F backward
    npts, F = 32 * 600 * 4, 32 * 6 * 14960 * 64 * 4
    assert abs(r["step_alg_GBps"] * r["ms_per_step"] * 1e6 - (npts * (4380 + 8504) + F)) <= 1e-6 * F
    alg = {n: v["alg_bytes"] for n, v in r["kernels"].items() if "alg_bytes" in v}
    assert alg["msmv_bwd_owner_dfeat"] + alg["msmv_bwd_gather_c64"] == npts * 8504 + F
    cpu = r["cpu_baseline"]
    assert cpu["kind"] in ("port", "reference") and cpu["cores"] >= 1 and cpu["value"] > 0
    assert cpu["unit"] == r["unit"] and cpu["sample"]


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["deferred", "accumulate"])
def test_decoder_workload_runs_in_both_dfeat_modes(mode):
    r = run_bench("--workload", "decoder-T", "--steps", "3", "--warmup", "1", "--no-cpu-baseline",
                  "--dfeat", mode)
    assert r["config"]["dfeat_sum"] == mode
    names = set(r["kernels"])
    if mode == "deferred":
        assert "msmv_bwd_owner_dfeat(6 layers)" in names and "msmv_bwd_owner_dfeat" not in names
    else:
        assert {"msmv_bwd_owner_dfeat", "msmv_bwd_owner_dfeat(accumulate)"} <= names
    assert r["config"]["points_per_step_per_gpu"] == 6 * 32 * 600 * 4


@pytest.mark.gpu
def test_training_step_workload_runs_on_one_gpu():
    """OPUS-M training-step shape on one rank: six layers x (2 forwards + 1 backward); the
    gradient all-reduce has no peer at N = 1 (its two-rank path runs under gloo in
    tests/test_dist_gloo.py)."""
    r = run_bench("--workload", "train-M", "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--event-steps", "3")
    assert r["config"]["points_per_step_per_gpu"] == 6 * 32 * 2400 * 2
    assert r["kernels"]["msmv_fwd_c64"]["launches"] == 3 * 12   # 6 layers x 2 forwards in each evented step
    assert r["value"] > 0


@pytest.mark.gpu
def test_rccl_calls_of_the_multi_rank_path_in_a_group_of_one():
    """The calls the N > 1 path makes - process-group initialisation bound to the device, barriers,
    the MAX reduction of the elapsed time, train-M's bucketed all-reduce on a side stream - through
    RCCL itself, in a group of one rank (this box has one GPU; RCCL refuses two ranks on one device)."""
    r = run_bench("--gpus", "1", "--init-group", "--workload", "train-M", "--steps", "2", "--warmup", "1",
                  "--settle-ms", "20", "--no-cpu-baseline", "--event-steps", "2", timeout=200)
    assert r["n_gpus"] == 1 and r["config"]["backend"] == "nccl"
    assert r["config"]["allreduce_bytes_per_step"] == 300 * 1000 * 1000
    # two buckets handed to RCCL after each layer's backward, one join at the end of the step
    assert r["config"]["allreduce_schedule"] == [[layer, 2 * layer + k] for layer in range(6) for k in range(2)]
    assert r["allreduce_exposed_us"] is not None
    assert r["value"] > 0


@pytest.mark.gpu
def test_self_launch_with_one_gpu_is_the_plain_run():
    r = run_bench("--gpus", "1", "--workload", "T-fwd", "--steps", "3", "--warmup", "1", "--no-cpu-baseline")
    assert r["n_gpus"] == 1 and r["config"]["backend"] is None
    assert r["config"]["ranks"] == [{"rank": 0, "seed": 1234, "slabs": [0, 32], "device": "cuda:0"}]


@pytest.mark.gpu
def test_build_then_smoke_in_one_process():
    """The driver's two hooks back to back in a fresh interpreter: the library must end up on the
    same HIP runtime as torch whichever of them is imported first."""
    code = "import __graft_entry__ as g; g.build(); g.smoke()"
    p = subprocess.run([sys.executable, "-c", code], cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    assert "smoke ok" in p.stdout


@pytest.mark.gpu
@pytest.mark.parametrize("workload", ["T-fwdbwd", "train-M"])
def test_two_real_ranks_on_one_device(workload):
    """The multi-rank path with real kernels: two ranks started by bench.py itself, both on this
    box's one GPU, gloo for the rendezvous, barriers and the slowest rank's time - and, for train-M,
    for the bucketed gradient all-reduce issued layer by layer.  gloo is given HOST tensors only
    (reducing device tensors of a GPU that two ranks share hung once in three runs), and the
    launcher ends ranks that are still running at its deadline."""
    r = run_bench("--gpus", "2", "--backend", "gloo", "--share-device", "--steps", "3", "--warmup", "1",
                  "--settle-ms", "20", "--workload", workload, "--no-cpu-baseline", "--event-steps", "2",
                  "--rank-timeout", "80", timeout=100)
    if workload == "train-M":
        assert r["config"]["allreduce_schedule"] == [[layer, 2 * layer + k] for layer in range(6) for k in range(2)]
        assert r["allreduce_exposed_us"] is not None
    assert r["n_gpus"] == 2 and r["config"]["shared_device"] is True and r["config"]["backend"] == "gloo"
    ranks = r["config"]["ranks"]
    assert [x["rank"] for x in ranks] == [0, 1] and ranks[0]["seed"] != ranks[1]["seed"]
    assert ranks[0]["slabs"] == [0, 32] and ranks[1]["slabs"] == [32, 64]
    per_rank = r["config"]["points_per_step_per_gpu"]
    assert abs(r["value"] - 2 * per_rank / (r["ms_per_step"] * 1e-3)) <= 1e-6 * r["value"]
    assert "cpu_baseline" not in r
