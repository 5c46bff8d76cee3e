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
    return json.loads(lines[0])


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
    assert roof["achieved"] > 0 and abs(roof["frac"] - roof["achieved"] / roof["peak"]) < 1e-9
    assert "traffic" in roof
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
def test_build_then_smoke_in_one_process():
    """The driver's two hooks back to back in a fresh interpreter: the library must end up on the
    same HIP runtime as torch whichever of them is imported first."""
    code = "import __graft_entry__ as g; g.build(); g.smoke()"
    p = subprocess.run([sys.executable, "-c", code], cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    assert "smoke ok" in p.stdout
