"""bench.py on the GPU box, at a small configuration: the JSON line's contract (one line on
stdout, the driver's keys, roofline + cpu_baseline objects), the counters measured by the
invocation's own rocprofv3 child runs, and the sharded product object as the thing measured."""

import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SMALL = ["--envs-per-gpu", "64", "--frame", "64", "--spp", "4", "--steps", "4", "--warmup", "1"]


def run_bench(*args):
    out = subprocess.run([sys.executable, "bench.py", *SMALL, *args], cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, out.stdout
    return json.loads(lines[0])


def test_bench_line_contract_with_counters_measured_in_the_invocation():
    line = run_bench("--cpu-baseline-envs", "64")
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in line, key
    assert line["n_gpus"] == 1 and line["steps"] == 4 and line["warmup"] == 1 and line["unit"] == "env-steps/s"
    assert line["value"] > 0 and abs(line["value"] - 64 * 4 / (line["ms_per_step"] * 4 / 1000.0)) < 1e-6 * line["value"]
    assert line["scaling"] == "weak" and line["vs_baseline"] is None and line["data"] == "synthetic"
    assert "workload" in line["config"] and "model" not in line["config"]
    roofline = line["roofline"]
    assert roofline["bound"] == "hbm" and roofline["peak"] == 8000.0 and roofline["unit"] == "GB/s"
    assert abs(roofline["frac"] - roofline["achieved"] / roofline["peak"]) < 1e-12
    assert roofline["kernel"].startswith("render_kernel_coop2<") and roofline["launches"] >= 4
    # measured by this invocation's rocprofv3 --pmc child runs (no committed profile exists for 64 px)
    assert roofline.get("pmc_failure") is None, roofline.get("pmc_failure")
    assert roofline["traffic_from_committed_profile"] is False and roofline["traffic"] > 0
    valu = line["roofline_valu"]
    assert valu["from_committed_profile"] is False and 0 < valu["frac"] < 1 and 0 < valu["lane_utilisation"] <= 1
    cpu = line["cpu_baseline"]
    assert cpu["kind"] == "port" and cpu["value"] > 0 and cpu["cores"] >= 1 and "64 envs" in cpu["sample"]
    assert str(cpu["cores"]) in cpu["thread_sweep_env_steps_per_s"] and len(cpu["other_configs"]) == 2


def test_bench_of_the_sharded_product_object():
    line = run_bench("--sharded-env", "--gpus", "1", "--no-cpu-baseline", "--no-pmc")
    assert line["n_gpus"] == 1 and line["value"] > 0
    assert "ShardedVectorDiscreteSteps" in line["config"]["sharding"]
    assert line["roofline"]["traffic_from_committed_profile"] is False and line["roofline"]["traffic"] is None
