"""The N > 1 path of bench.py on CPU: two ranks over gloo -- launched by bench.py itself
(`python bench.py --gpus 2`, what the driver runs) and by torch.distributed.run --, env sharding
plan, barrier + max-over-ranks timing, one JSON line from rank 0, a failing rank failing the
launch.  No GPU work happens in --plumbing-test mode and the line says so."""

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
    lines = [l for l in out.stdout.splitlines() if l.strip()]
    assert len(lines) == 1 and lines[0].startswith("{"), out.stdout  # stdout is the JSON line and nothing else
    return json.loads(lines[0])


def test_single_rank_plumbing_line():
    line = _run([sys.executable, "bench.py", "--plumbing-test", "--steps", "3", "--warmup", "1"])
    assert line["n_gpus"] == 1 and line["steps"] == 3 and line["warmup"] == 1
    assert line["value"] is None and "plumbing" in line["data"]
    assert line["scaling"] == "weak" and line["higher_is_better"] is True
    assert line["config"]["host_gather_bytes_per_step"] == 29 * line["config"]["total_envs"]  # (one rank: its own block)


def test_two_ranks_over_gloo():
    line = _run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                 "--master-addr", "127.0.0.1", "--master-port", "29517", "bench.py", "--gpus", "2",
                 "--plumbing-test", "--steps", "4", "--warmup", "1"])
    assert line["n_gpus"] == 2
    assert line["config"]["total_envs"] == 2 * line["config"]["envs_per_gpu"]
    assert line["ms_per_step"] > 0
    # every step's block of every rank (29 B per environment) is gathered on rank 0 inside the timed region: one vector
    # environment over all ranges returns ONE obs[N, 4] (vector_environment.py:104-164)
    assert line["config"]["host_gather_bytes_per_step"] == 29 * line["config"]["total_envs"]
    assert "gathered on rank 0" in line["config"]["sharding"]


def test_eight_ranks_over_gloo_with_eight_distinct_gpus():
    """BASELINE.json configs[3] as the driver launches it -- 8 ranks x 4096 environments -- without GPUs: the self-launcher,
    the rendezvous, every step's 32 768 x 29 bytes gathered on rank 0, the table of eight distinct (fake) PCI bus ids in rank
    order, every rank's own time per step, and the launcher's wall time (eight interpreters with torch on this host's
    eight cores)."""
    import time

    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    buses = ["0000:%02x:00.0@%d" % (0x05 + 0x20 * g, g // 4) for g in range(8)]
    env["REINFOCUS_BENCH_FAKE_DEVICES"] = ",".join(buses)
    started = time.monotonic()
    out = subprocess.run([sys.executable, "bench.py", "--gpus", "8", "--plumbing-test", "--steps", "5", "--warmup", "1"],
                         cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    wall = time.monotonic() - started
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.strip()]
    assert len(lines) == 1 and lines[0].startswith("{"), out.stdout
    line = json.loads(lines[0])
    assert line["n_gpus"] == 8 and line["config"]["envs_per_gpu"] == 4096 and line["config"]["total_envs"] == 32768
    assert line["config"]["host_gather_bytes_per_step"] == 29 * 32768
    assert [(d["rank"], d["pci_bus_id"], d["numa_node"]) for d in line["devices"]] == \
        [(g, buses[g].split("@")[0], g // 4) for g in range(8)]
    assert len(line["per_rank_ms_per_step"]) == 8 and all(ms > 0 for ms in line["per_rank_ms_per_step"])
    assert max(line["per_rank_ms_per_step"]) <= line["ms_per_step"] * 1.001  # the line's time is the slowest rank's
    assert wall < 300, wall
    # seven distinct GPUs for eight ranks is not an 8-GPU measurement
    env["REINFOCUS_BENCH_FAKE_DEVICES"] = ",".join(buses[:7] + [buses[3]])
    out = subprocess.run([sys.executable, "bench.py", "--gpus", "8", "--plumbing-test", "--steps", "1", "--warmup", "0"],
                         cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode != 0 and "both ran on the GPU at " + buses[3].split("@")[0] in out.stderr


def test_bench_launches_its_own_ranks():
    """`python bench.py --gpus 2 ...` with no launcher around it: the parent starts the two ranks
    (before anything could touch a GPU), relays rank 0's line and exits 0."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    out = subprocess.run([sys.executable, "bench.py", "--gpus", "2", "--plumbing-test", "--steps", "4", "--warmup", "1"],
                         cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.strip()]
    assert len(lines) == 1 and lines[0].startswith("{"), out.stdout  # stdout is the JSON line and nothing else
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["steps"] == 4
    assert line["config"]["total_envs"] == 2 * line["config"]["envs_per_gpu"]
    assert "one process per GPU" in line["config"]["sharding"]


def test_the_line_names_one_distinct_gpu_per_rank():
    """Every rank reports which physical GPU it used (PCI bus id, NUMA node) and the CPUs it ran on; rank 0 prints
    the table and the launch fails when two ranks name the same GPU (here from a fake table: no GPU on this host)."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    env["REINFOCUS_BENCH_FAKE_DEVICES"] = "0000:05:00.0@0,0000:c5:00.0@1"
    cmd = [sys.executable, "bench.py", "--gpus", "2", "--plumbing-test", "--steps", "2", "--warmup", "0"]
    out = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    line = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][0])
    assert [(d["rank"], d["device"], d["pci_bus_id"], d["numa_node"]) for d in line["devices"]] == \
        [(0, 0, "0000:05:00.0", 0), (1, 1, "0000:c5:00.0", 1)]
    assert all(d["cpus"] and d["host"] for d in line["devices"])
    # two ranks on one GPU: not a 2-GPU measurement
    env["REINFOCUS_BENCH_FAKE_DEVICES"] = "0000:05:00.0@0"
    out = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode != 0 and "both ran on the GPU at 0000:05:00.0" in out.stderr
    assert not [l for l in out.stdout.splitlines() if l.startswith("{")]
    # ... unless the run says it is a rehearsal on one device
    env["REINFOCUS_BENCH_DEVICE"] = "0"
    out = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]


def test_a_failing_rank_fails_the_launch():
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    env["REINFOCUS_BENCH_FAIL_RANK"] = "1"
    out = subprocess.run([sys.executable, "bench.py", "--gpus", "2", "--plumbing-test", "--steps", "2", "--warmup", "0"],
                         cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 3, (out.returncode, out.stderr[-2000:])
    assert "rank 1 exited with code 3" in out.stderr
    assert not [l for l in out.stdout.splitlines() if l.startswith("{")]


def _children_of(pid):
    try:
        return [int(c) for c in open(f"/proc/{pid}/task/{pid}/children").read().split()]
    except OSError:
        return []


def _alive(pid):
    try:
        return open(f"/proc/{pid}/stat").read().rsplit(")", 1)[1].split()[0] != "Z"
    except OSError:
        return False


def test_a_terminated_launcher_takes_its_ranks_with_it():
    """The driver may end `python bench.py --gpus N` with SIGTERM (a timeout): the launcher must not
    leave ranks behind holding GPUs.  It ends them by PID and exits with 128 + SIGTERM."""
    import signal
    import time

    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    parent = subprocess.Popen([sys.executable, "bench.py", "--gpus", "2", "--plumbing-test", "--steps", "100000",
                               "--warmup", "0"], cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    try:
        deadline = time.monotonic() + 120
        children = []
        while len(children) < 2 and time.monotonic() < deadline:
            children = _children_of(parent.pid)
            time.sleep(0.1)
        assert len(children) == 2, children
        time.sleep(1.0)
        parent.send_signal(signal.SIGTERM)
        assert parent.wait(timeout=60) == 128 + signal.SIGTERM
        deadline = time.monotonic() + 20
        while any(_alive(c) for c in children) and time.monotonic() < deadline:
            time.sleep(0.1)
        assert not any(_alive(c) for c in children), "ranks survived their launcher"
    finally:
        if parent.poll() is None:
            parent.kill()


def test_a_rank_that_ignores_sigterm_is_killed():
    """Rank 1 dies before the rendezvous, rank 0 hangs and ignores SIGTERM: the launcher waits 10 s,
    terminates, and after the grace period kills -- it must not wait for ever."""
    import time

    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    env.update(REINFOCUS_BENCH_FAIL_EARLY_RANK="1", REINFOCUS_BENCH_HANG_RANK="0")
    t0 = time.monotonic()
    out = subprocess.run([sys.executable, "bench.py", "--gpus", "2", "--plumbing-test", "--steps", "2", "--warmup", "0"],
                         cwd=ROOT, env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 3, (out.returncode, out.stderr[-2000:])
    assert "ending ranks [0]" in out.stderr
    assert time.monotonic() - t0 < 120


def test_parent_launcher_never_loads_the_hip_library():
    """The launching parent must not initialise the GPU (a process that has may not hand over to
    others on this pool): launch_ranks is reached before reinfocus_amd._native is imported."""
    import ast

    tree = ast.parse(open(os.path.join(ROOT, "bench.py")).read())
    main = next(n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name == "main")
    launch_line = next(n.lineno for n in ast.walk(main) if isinstance(n, ast.Call)
                       and getattr(n.func, "id", "") == "launch_ranks")
    native_lines = [n.lineno for n in ast.walk(main) if isinstance(n, ast.ImportFrom) and n.module
                    and n.module.startswith("reinfocus_amd")]
    assert native_lines and launch_line < min(native_lines)
    top_level = [n for n in tree.body if isinstance(n, (ast.Import, ast.ImportFrom))]
    assert not any("reinfocus_amd" in ast.dump(n) or "torch" in ast.dump(n) for n in top_level)


def test_committed_profile_is_matched_by_kernel_and_configuration():
    """roofline.traffic / roofline_valu come from the committed PMC summary of exactly the kernel
    instance, frame size and sample count of the run -- and from nothing else."""
    sys.path.insert(0, ROOT)
    import bench

    assert bench.kernel_key("void rf::render_kernel_coop2<true, 1, 4, 32, false>(rf::RenderArgs)") == \
        bench.kernel_key("render_kernel_coop2<true, 1, 4, 32>") == "render_kernel_coop2<true,1,4,32>"
    assert bench.kernel_key("void rf::render_kernel_coop2<true, 1, 4, 32, true>(rf::RenderArgs)") == \
        "render_kernel_coop2<true,1,4,32,true>"  # the two-pass instance of the fused env step
    fused, why_not = bench.committed_profile("render_kernel_coop2<true, 1, 4, 32, true>", 256, 16)
    assert fused is not None, why_not
    # (per pixel really rendered -- second passes included --, 16 samples each: ~263 instructions per 64 pixel-samples)
    assert 230 < bench.roofline_from_profile(fused, 4096 * 65536)[1]["valu_insts_per_64_pixels"] / 16 < 300
    profile, why_not = bench.committed_profile("render_kernel_coop2<true, 1, 4, 32>", 256, 16)
    assert profile is not None, why_not
    assert profile["config"]["frame"] == 256 and profile["config"]["spp"] == 16 and profile["commit"]
    traffic, valu = bench.roofline_from_profile(profile, 4096 * 65536)
    assert 0.9 < traffic / (35 * 4096 * 65536) < 1.1          # HBM traffic ~ the algorithmic bytes
    assert valu["from_committed_profile"] is True and 0.3 < valu["frac"] < 1.0
    assert 0 < valu["barrier_wait_share"] < 1 and 0 < valu["salu_per_valu"] < 1
    # another kernel instance, or another configuration: no figures rather than wrong ones
    assert bench.committed_profile("render_kernel_coop2<true, 0, 4, 32>", 256, 16)[0] is None
    assert bench.committed_profile("render_kernel_coop2<true, 1, 4, 32>", 256, 32)[0] is None
    ref300, _ = bench.committed_profile("render_kernel_coop2<false, 1, 2, 32>", 300, 100)
    assert ref300 is not None and "_ref300" in ref300["file"]  # the newest round's summary of that configuration
    # what issue rate this kernel's instruction mix can reach stands next to the generic peak of 0.5
    low, high = valu["attainable_for_this_mix"]["no_overlap"], valu["attainable_for_this_mix"]["fast_fp_hidden"]
    assert 0.2 < low < high < 0.5 and abs(sum(valu["attainable_for_this_mix"]["shares"].values()) - 1) < 1e-9
    assert valu["frac_of_attainable"][0] < valu["frac_of_attainable"][1]
