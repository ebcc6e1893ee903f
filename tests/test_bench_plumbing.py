"""The N > 1 path of bench.py on CPU: two ranks over gloo (torch.distributed.run,
127.0.0.1), env sharding plan, barrier + max-over-ranks timing, one JSON line from rank 0.
No GPU work happens in --plumbing-test mode and the line says so."""

import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_shard_plan_partitions_the_state_sequence():
    sys.path.insert(0, ROOT)
    import bench

    plans = [bench.shard_plan(r, 4096, 256) for r in range(8)]
    assert [p["first_env"] for p in plans] == [r * 4096 for r in range(8)]
    for r in range(7):
        # contiguous, non-overlapping slices of the global pixel/state index space
        assert plans[r + 1]["first_state_index"] - plans[r]["first_state_index"] == 4096 * 256 * 256
    assert plans[7]["first_state_index"] + 4096 * 256 * 256 == 32768 * 256 * 256  # BASELINE configs[3]


def _run(cmd):
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    out = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout
    return json.loads(lines[0])


def test_single_rank_plumbing_line():
    line = _run([sys.executable, "bench.py", "--plumbing-test", "--steps", "3", "--warmup", "1"])
    assert line["n_gpus"] == 1 and line["steps"] == 3 and line["warmup"] == 1
    assert line["value"] is None and "plumbing" in line["data"]
    assert line["scaling"] == "weak" and line["higher_is_better"] is True


def test_two_ranks_over_gloo():
    line = _run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                 "--master-addr", "127.0.0.1", "--master-port", "29517", "bench.py", "--gpus", "2",
                 "--plumbing-test", "--steps", "4", "--warmup", "1"])
    assert line["n_gpus"] == 2
    assert line["config"]["total_envs"] == 2 * line["config"]["envs_per_gpu"]
    assert line["ms_per_step"] > 0
