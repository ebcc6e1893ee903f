"""Headline benchmark: vectorised env-steps/s of DiscreteSteps-v0 on MI355X.

    python bench.py --gpus N --steps K --warmup W

With N > 1 and no WORLD_SIZE in the environment this process only LAUNCHES: before anything
touches the GPU it starts N fresh children of itself (RANK / LOCAL_RANK / WORLD_SIZE /
MASTER_ADDR=127.0.0.1 / MASTER_PORT set), relays rank 0's JSON line and exits with the worst
child's code.  Under an external launcher (python -m torch.distributed.run --nproc-per-node N
bench.py --gpus N ...) it is one of the ranks.

A "step" is one VectorDiscreteSteps.step(actions) over this rank's environments: transform /
enders / rewards, scene packing, the render kernel, the focus kernel, an 8 B/env D2H and the
same-step partial render of the environments that just ended.
Workload = BASELINE.json configs[2]: 4096 envs x 256x256 x 16 spp per GPU; with N > 1 every rank
owns 4096 more envs (weak scaling, BASELINE configs[3]) and its own slice of the RNG state
sequence.  There is no data-path collective: environments are independent.  What the one vector
environment the ranks stand for returns from step() -- ONE obs[N, 4], rewards[N], flags[N]
(vector_environment.py:104-164) -- is put together on the host: every step, inside the timed
region, rank 0 gathers the ranks' blocks (29 B/env) over gloo; otherwise the ranks meet only at the
timing barriers.  --sharded-env measures the product-level object instead
(harness.ShardedVectorDiscreteSteps: one process, one rf_ctx + one host thread per GPU, step()
concatenates).

Prints ONE JSON line (rank 0).  value = all ranks' env-steps / max-over-ranks seconds, with
inputs resident in HBM (the only per-step host traffic is 4 + 8 B/env of actions and candidate
reset states in and 29 B/env of observations, rewards and flags out, inside the timed region).
"""

import argparse
import json
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
RENDER_BYTES_PER_PIXEL = 35  # 16 B state read + 16 B state write + 3 B frame write
FOCUS_BYTES_PER_PIXEL = 3    # frame read
VALU_ISSUE_PEAK = 0.5        # wave64 VALU instructions per cycle per SIMD (2 cycles each), same guide


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--envs-per-gpu", type=int, default=4096)
    ap.add_argument("--frame", type=int, default=256)
    ap.add_argument("--spp", type=int, default=16)
    ap.add_argument("--cpu-baseline-envs", type=int, default=0, help="0 = choose ~12 s of CPU work")
    ap.add_argument("--env", choices=["device", "host", "literal"], default="device",
                    help="device: whole step resident on the GPU (rf_env_*); host: numpy glue around "
                         "rf_render / rf_focus on frames that stay in HBM; literal: the same glue over "
                         "INTEGRATION.md's literal stub -- rf_render(host_out) -> numpy array -> rf_upload_frames + "
                         "rf_focus, i.e. every frame crosses PCIe twice per render (identical results, "
                         "tests/test_gpu_environment.py)")
    ap.add_argument("--sharded-env", action="store_true",
                    help="one process: harness.ShardedVectorDiscreteSteps over --gpus devices (threads), instead "
                         "of one process per GPU")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-pmc", action="store_true",
                    help="take roofline.traffic and roofline_valu from the committed profile instead of measuring "
                         "them in this invocation (by default, at N = 1: four short child runs of this script under "
                         "rocprofv3 --pmc after the timed run -- counters only, one group per run; about a minute)")
    ap.add_argument("--no-kernel-timing", action="store_true",
                    help="no per-kernel HIP events (and no roofline object): lets small configurations replay "
                         "their step as one hipGraph")
    ap.add_argument("--plumbing-test", action="store_true",
                    help="no GPU work: exercises only the multi-rank plumbing (CPU tests)")
    return ap.parse_args(argv)


# ---------------------------------------------------------------------------------------------
# launching the ranks (parent process: nothing here may touch the GPU)
# ---------------------------------------------------------------------------------------------
def free_port():
    import socket

    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def launch_ranks(world, argv, total_timeout=None):
    """Starts `world` children of this script, one rank per GPU, relays rank 0's stdout (the JSON
    line) and returns the worst exit code.  A rank that fails takes the others down with it (they
    would wait for it at the next barrier).  Children are only ever ended by PID: terminate(), then
    kill() after a grace period; the same happens when this launcher is interrupted or terminated
    (it never touches the GPU, so handling the signal here is safe) and when `total_timeout`
    seconds (REINFOCUS_BENCH_LAUNCH_TIMEOUT, default 3000) pass."""
    import signal
    import threading

    if total_timeout is None:
        total_timeout = float(os.environ.get("REINFOCUS_BENCH_LAUNCH_TIMEOUT", "3000"))
    port = free_port()
    children = []

    def end_children(grace=5.0):
        alive = [c for c in children if c.poll() is None]
        for child in alive:
            child.terminate()
        deadline = time.monotonic() + grace
        for child in alive:
            try:
                child.wait(timeout=max(0.0, deadline - time.monotonic()))
            except subprocess.TimeoutExpired:
                child.kill()
        for child in alive:
            try:
                child.wait(timeout=5.0)
            except subprocess.TimeoutExpired:
                pass

    interrupted = []  # signals received: the handler only records them, the poll loop acts (an exception raised
                      # from the handler could fire inside the clean-up itself and leave ranks behind)

    def on_signal(signum, frame):
        interrupted.append(signum)

    previous = {}
    for signum in (signal.SIGTERM, signal.SIGINT):
        try:
            previous[signum] = signal.signal(signum, on_signal)
        except ValueError:  # not the main thread (tests): the finally clause still cleans up
            pass
    worst = 0
    try:
        for rank in range(world):
            env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world),
                       MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
            env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
            # as torch.distributed.run does: N ranks with an OpenMP team of every core each (torch's, behind the per-step gloo
            # gather) starve each other and the HIP runtime's threads -- 60 instead of 25.5 ms per rank and step in a 6-rank
            # rehearsal (profiles/r05_ab.txt section 9)
            env.setdefault("OMP_NUM_THREADS", "1")
            out = subprocess.PIPE if rank == 0 else sys.stderr
            children.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + list(argv), env=env,
                                             stdout=out, cwd=os.getcwd()))

        def relay(pipe):  # rank 0's JSON line goes to stdout, anything else it printed to stderr
            for line in iter(pipe.readline, b""):
                text = line.decode("utf-8", "replace")
                out = sys.stdout if text.lstrip().startswith("{") else sys.stderr
                out.write(text)
                out.flush()

        pump = threading.Thread(target=relay, args=(children[0].stdout,), daemon=True)
        pump.start()
        started = time.monotonic()
        pending = set(range(world))
        failed_at = None
        while pending:
            if interrupted:
                print(f"bench.py: launcher got signal {interrupted[0]}, ending the ranks", file=sys.stderr)
                worst = 128 + int(interrupted[0])
                break
            for rank in sorted(pending):
                rc = children[rank].poll()
                if rc is None:
                    continue
                pending.discard(rank)
                if rc != 0:
                    print(f"bench.py: rank {rank} exited with code {rc}", file=sys.stderr)
                    worst = rc if worst == 0 or abs(rc) > abs(worst) else worst
                    failed_at = failed_at or time.monotonic()
            if pending and failed_at is not None and time.monotonic() - failed_at > 10.0:
                print(f"bench.py: ending ranks {sorted(pending)} (stuck behind a failed rank)", file=sys.stderr)
                end_children()  # the survivors are stuck at a barrier
                pending.clear()
            elif pending and time.monotonic() - started > total_timeout:
                print(f"bench.py: ranks {sorted(pending)} still running after {total_timeout:.0f} s", file=sys.stderr)
                worst = worst or 124
                end_children()
                pending.clear()
            time.sleep(0.05)
        if not interrupted:
            pump.join(timeout=5.0)
    finally:
        end_children()  # (further signals only append to `interrupted`: nothing can cut this short)
        for signum, handler in previous.items():
            signal.signal(signum, handler)
    return worst if 0 <= worst < 256 else 1


class Ranks:
    """Host-side rendezvous of the one-process-per-GPU ranks (gloo; no GPU tensors)."""

    def __init__(self, gpus, single_process=False):
        self.rank = int(os.environ.get("RANK", "0"))
        self.local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        self.world = 1 if single_process else int(os.environ.get("WORLD_SIZE", "1"))
        if self.world != gpus and not single_process:
            raise SystemExit(f"--gpus {gpus} but WORLD_SIZE={self.world}")
        self.dist = None
        if self.world > 1:
            import torch.distributed as dist

            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            # gloo announces its connections on stdout, which belongs to the one JSON line:
            # file descriptor 1 points at stderr while the group forms
            sys.stdout.flush()
            saved = os.dup(1)
            os.dup2(2, 1)
            try:
                dist.init_process_group(backend="gloo", rank=self.rank, world_size=self.world)
                dist.barrier()
            finally:
                sys.stdout.flush()
                os.dup2(saved, 1)
                os.close(saved)
            self.dist = dist

    def barrier(self):
        if self.dist is not None:
            self.dist.barrier()

    def reduce(self, value, op):
        if self.dist is None:
            return float(value)
        import torch

        t = torch.tensor([float(value)], dtype=torch.float64)
        self.dist.all_reduce(t, op=getattr(self.dist.ReduceOp, op))
        return float(t.item())

    def gather(self, item):
        """Every rank's `item` (any picklable object), in rank order, on every rank."""
        if self.dist is None:
            return [item]
        items = [None] * self.world
        self.dist.all_gather_object(items, item)
        return items

    STEP_BYTES_PER_ENV = 29  # observations float32 x 4, reward float64, terminated, truncated, ... : harness's per-env output

    def gather_step(self, obs, rewards, terminated, truncated):
        """What one environment over all ranks' ranges returns from step() is ONE obs[N, 4] / rewards[N] / flags[N]
        (vector_environment.py:104-164): rank 0 receives every rank's block -- 29 B per environment over gloo, the
        host-side gather north_star allows -- inside the timed region.  Returns the bytes rank 0 holds afterwards."""
        n = len(rewards)
        if self.dist is None:
            return n * self.STEP_BYTES_PER_ENV
        import torch

        block = np.empty((n, self.STEP_BYTES_PER_ENV), dtype=np.uint8)
        block[:, 0:16] = np.ascontiguousarray(obs, dtype=np.float32).reshape(n, 4).view(np.uint8)
        block[:, 16:24] = np.ascontiguousarray(rewards, dtype=np.float64).reshape(n, 1).view(np.uint8)
        block[:, 24] = terminated
        block[:, 25] = truncated
        block[:, 26:] = 0
        mine = torch.from_numpy(block)
        if self.rank == 0:
            if getattr(self, "_gathered", None) is None or self._gathered[0].shape != mine.shape:
                self._gathered = [torch.empty_like(mine) for _ in range(self.world)]
            self.dist.gather(mine, self._gathered, dst=0)
            return sum(int(t.numel()) for t in self._gathered)
        self.dist.gather(mine, None, dst=0)
        return 0

    def close(self):
        if self.dist is not None:
            self.dist.destroy_process_group()


# ---------------------------------------------------------------------------------------------
# figures taken from the committed rocprofv3 PMC summary (profiles/<tag>_pmc.json)
# ---------------------------------------------------------------------------------------------
def kernel_key(name):
    """'void rf::render_kernel_coop2<true, 1, 4, 32, false>(rf::RenderArgs)' -> 'render_kernel_coop2<true,1,4,32>'
    (the profiler prints the defaulted last template argument -- false: single pass --, the library's own name for
    the instance leaves it out; the two-pass instance of the fused env step keeps its ',true')."""
    name = name.split("(")[0].strip()
    if name.startswith("void "):
        name = name[5:]
    name = name.replace("rf::", "").replace(" ", "")
    if name.startswith("render_kernel_coop2<") and name.endswith(",false>"):
        name = name[:-len(",false>")] + ">"
    return name


def committed_profile(kernel, frame, spp):
    """The newest committed PMC summary (by round tag) that was collected for exactly this kernel
    instance at this frame size and sample count, or (None, reason).  These figures are NOT measured
    by the run that prints them; they are marked as such wherever they appear."""
    import glob

    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc.json")), reverse=True)
    if not files:
        return None, "no profiles/r*_pmc.json"
    for path in files:
        data = json.load(open(path))
        meta = data.get("_meta", {})
        config = meta.get("config", {})
        if not config or config.get("frame") != frame or config.get("spp") != spp:
            continue
        for name, entry in data.items():
            if kernel_key(name) == kernel_key(kernel) and entry.get("SQ_WAVES"):
                return {"file": os.path.basename(path), "commit": meta.get("commit"), "config": config,
                        "entry": entry}, None
    return None, f"no committed PMC summary for {kernel} at frame {frame} / spp {spp}"


def committed_static_mix(kernel):
    """The static VALU mix by gfx950 issue class that profiles/summarize.py stored for this kernel instance
    (newest round first), or None: what issue rate this instruction mix can reach (see DESIGN.md 4.1 "Roofline")."""
    import glob

    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc.json")), reverse=True):
        meta = json.load(open(path)).get("_meta", {})
        if kernel_key((meta.get("config") or {}).get("kernel") or "") == kernel_key(kernel) and meta.get("static_valu_mix"):
            return dict(meta["static_valu_mix"], file=os.path.basename(path), commit=meta.get("commit"))
    return None


def roofline_from_profile(profile, pixels_per_launch):
    e = profile["entry"]
    pixels = e["SQ_WAVES"] * float(e.get("pixels_per_wave", 64))
    traffic = None
    if "hbm_read_bytes_corrected" in e and "hbm_write_bytes" in e:
        traffic = (e["hbm_read_bytes_corrected"] + e["hbm_write_bytes"]) / pixels * pixels_per_launch
    valu = None
    if e.get("valu_insts_per_cycle_per_simd") is not None:
        valu = {
            "bound": "valu-issue",
            "achieved": e["valu_insts_per_cycle_per_simd"],
            "peak": VALU_ISSUE_PEAK,
            "unit": "wave64 VALU instructions / cycle / SIMD",
            "frac": e["valu_insts_per_cycle_per_simd"] / VALU_ISSUE_PEAK,
            "formula": "SQ_INSTS_VALU / (GRBM_GUI_ACTIVE / 8 XCDs * 1024 SIMDs), peak = 1 instruction per 2 cycles",
            "lane_utilisation": e.get("valu_lane_utilisation"),
            "lane_ops_frac_of_peak": (e["valu_insts_per_cycle_per_simd"] / VALU_ISSUE_PEAK
                                      * e.get("valu_lane_utilisation", 1.0)),
            "barrier_wait_share": e.get("wait_share"),        # SQ_WAIT_ANY / SQ_WAVE_CYCLES
            "inst_wait_share": e.get("inst_wait_share"),      # SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES
            "salu_per_valu": e.get("salu_per_valu"),          # SQ_INSTS_SALU / SQ_INSTS_VALU
            "valu_insts_per_64_pixels": e["valu_insts_per_wave"] * 64.0 / float(e.get("pixels_per_wave", 64)),
            "pixels_from": e.get("pixels_from", "SQ_WAVES x pixels per wave (padded lanes included)"),
            "from_committed_profile": profile["file"] is not None,
            "measured_by": ("profiles/" + profile["file"] if profile["file"] else
                            "rocprofv3 --pmc child runs of this invocation"),
            "profile": {k: profile[k] for k in ("file", "commit", "config")},
        }
        mix = committed_static_mix(profile["config"]["kernel"]) if profile.get("config") else None
        if mix:
            # 0.5 is reachable only by streams of fast-path instructions; this kernel's mix (static, by issue
            # class, with the per-class costs of tools/ubench/pairbench) can reach between the first figure
            # (nothing overlaps) and the second (the fast-path float operations hide behind slow-path ones)
            low, high = mix["attainable_insts_per_cycle_per_simd"]
            valu["attainable_for_this_mix"] = {"no_overlap": low, "fast_fp_hidden": high, "shares": mix["shares"],
                                               "cycles_per_class": mix["cycles_per_class"],
                                               "from_committed_profile": mix["file"], "commit": mix["commit"]}
            valu["frac_of_attainable"] = [valu["achieved"] / high, valu["achieved"] / low]
    return traffic, valu


PMC_GROUPS = (
    "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_INST_ANY",
    "SQ_WAIT_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE GRBM_COUNT",
    "FETCH_SIZE",
    "WRITE_SIZE",
)


def derive_pmc(e, kernel):
    """Derived figures of one kernel's PMC totals (the same definitions as profiles/summarize.py)."""
    e = dict(e)
    if "FETCH_SIZE" in e:
        e["hbm_read_bytes_corrected"] = e["FETCH_SIZE"] * 1024 * 2  # KiB; gfx950 counts wide reads at half
    if "WRITE_SIZE" in e:
        e["hbm_write_bytes"] = e["WRITE_SIZE"] * 1024
    # pixels: what the profiled process really rendered with this kernel (rf_pixels_rendered, "pixels" in
    # the totals); the fallback -- waves x pixels per wave -- counts the padded lanes of partial tiles as
    # pixels and is only right for frames that are a whole number of tiles and the default kSets = 3
    if e.get("pixels") and e.get("SQ_WAVES"):
        e["pixels_per_wave"] = e["pixels"] / e["SQ_WAVES"]
        e["pixels_from"] = "rf_pixels_rendered of the profiled run"
    else:
        e["pixels_per_wave"] = 192 if ("render_kernel_coop2" in kernel or "render_kernel_wave" in kernel) else 64  # (the wave kernel: K = 3)
        e["pixels_from"] = "SQ_WAVES x pixels per wave (padded lanes included)"
    if e.get("SQ_WAVES") and "SQ_INSTS_VALU" in e:
        e["valu_insts_per_wave"] = e["SQ_INSTS_VALU"] / e["SQ_WAVES"]
    if e.get("GRBM_GUI_ACTIVE") and e.get("SQ_INSTS_VALU"):
        e["valu_insts_per_cycle_per_simd"] = e["SQ_INSTS_VALU"] / (e["GRBM_GUI_ACTIVE"] / 8.0 * 1024.0)
    if e.get("SQ_ACTIVE_INST_VALU") and "SQ_THREAD_CYCLES_VALU" in e:
        e["valu_lane_utilisation"] = e["SQ_THREAD_CYCLES_VALU"] / (e["SQ_ACTIVE_INST_VALU"] * 64)
    if e.get("SQ_WAVE_CYCLES"):
        if "SQ_WAIT_ANY" in e:
            e["wait_share"] = e["SQ_WAIT_ANY"] / e["SQ_WAVE_CYCLES"]
        if "SQ_WAIT_INST_ANY" in e:
            e["inst_wait_share"] = e["SQ_WAIT_INST_ANY"] / e["SQ_WAVE_CYCLES"]
    if e.get("SQ_INSTS_VALU") and "SQ_INSTS_SALU" in e:
        e["salu_per_valu"] = e["SQ_INSTS_SALU"] / e["SQ_INSTS_VALU"]
    return e


def measure_pmc(args, kernel):
    """Counter totals of `kernel` from child runs of this script under rocprofv3 --pmc (one counter
    group per run, never combined with a tracing domain, program directly after `--`), as
    /opt/skills/guides/MI355X_MICROARCH.md prescribes.  Returns a profile dict like
    committed_profile()'s, or (None, reason)."""
    import csv
    import glob
    import shutil
    import tempfile

    rocprof = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(rocprof):
        return None, "rocprofv3 not found"
    work = tempfile.mkdtemp(prefix="reinfocus_pmc_", dir="/tmp")
    child = [sys.executable, os.path.abspath(__file__), "--steps", "3", "--warmup", "1", "--no-cpu-baseline", "--no-pmc",
             "--envs-per-gpu", str(args.envs_per_gpu), "--frame", str(args.frame), "--spp", str(args.spp),
             "--env", args.env]
    totals = {}
    try:
        for index, group in enumerate(PMC_GROUPS):
            out = os.path.join(work, f"pass{index}")
            cmd = [rocprof, "--pmc", *group.split(), "--output-format", "csv", "-d", out, "--", *child]
            done = subprocess.run(cmd, cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"), capture_output=True, text=True,
                                  timeout=240)
            files = glob.glob(os.path.join(out, "*", "*_counter_collection.csv"))
            if done.returncode != 0 or not files:
                return None, f"rocprofv3 pass {index} failed (rc {done.returncode}): {done.stderr[-300:]}"
            for row in csv.DictReader(open(files[0])):
                if kernel_key(row["Kernel_Name"]) == kernel_key(kernel):
                    totals[row["Counter_Name"]] = totals.get(row["Counter_Name"], 0.0) + float(row["Counter_Value"])
            lines = [l for l in done.stdout.splitlines() if l.startswith("{")]
            if lines and "pixels" not in totals:  # every pass renders the same pixels (same seeds)
                child_line = json.loads(lines[-1])
                if child_line.get("render_pixels_by_kernel", {}).get(kernel_key(kernel)):
                    totals["pixels"] = float(child_line["render_pixels_by_kernel"][kernel_key(kernel)])
        if not totals.get("SQ_WAVES"):
            return None, f"no counters for {kernel}"
        return {"file": None, "commit": None, "entry": derive_pmc(totals, kernel),
                "config": {"envs": args.envs_per_gpu, "frame": args.frame, "spp": args.spp, "kernel": kernel}}, None
    except (OSError, subprocess.TimeoutExpired) as error:
        return None, f"rocprofv3: {error}"
    finally:
        shutil.rmtree(work, ignore_errors=True)


def baseline_config_name(envs_per_gpu, frame, spp, n_gpus):
    """Which BASELINE.json configuration a run is (so that the line never claims another one's name)."""
    total = envs_per_gpu * n_gpus
    if (frame, spp) == (256, 16) and envs_per_gpu == 4096:
        return ("BASELINE.json configs[2]" if n_gpus == 1 else
                "BASELINE.json configs[3]" if n_gpus == 8 else
                f"BASELINE.json configs[2] per GPU, weak-scaled to {n_gpus} GPUs (configs[3] is the 8-GPU point)")
    if (frame, spp) == (512, 64) and total == 1024 and n_gpus == 8:
        return "BASELINE.json configs[4]"
    if (frame, spp) == (512, 64) and envs_per_gpu == 128:
        return f"per-GPU share of BASELINE.json configs[4] (1024 envs over 8 GPUs) on {n_gpus} GPU(s)"
    if (envs_per_gpu, frame, spp, n_gpus) == (256, 128, 4, 1):
        return "BASELINE.json configs[1]"
    if (envs_per_gpu, frame, spp, n_gpus) == (1, 64, 1, 1):
        return "BASELINE.json configs[0] on the GPU path"
    if (frame, spp) == (300, 100):
        return "not a BASELINE.json configuration: the reference's default frame size and sample count"
    return "not a BASELINE.json configuration"


def rank_placement(rank, device, native=None):
    """Where this rank runs: HIP device index, the GPU's PCI bus id and NUMA node (rf_device_info), and the CPUs the
    whole process -- the HIP runtime's helper threads included -- was restricted to: those of that NUMA node
    (REINFOCUS_BENCH_NO_PIN=1 leaves the affinity alone).  Without `native` (the plumbing test, no GPU) the row comes
    from REINFOCUS_BENCH_FAKE_DEVICES = "busid@node,busid@node,..." indexed by rank."""
    import socket

    if native is None:
        table = [entry.split("@") for entry in os.environ.get("REINFOCUS_BENCH_FAKE_DEVICES", "").split(",") if entry]
        bus, node = table[rank % len(table)] if table else (f"fake:{rank:02x}:00.0", "-1")
        info = {"device": device, "pci_bus_id": bus, "numa_node": int(node)}
        cpus = None
    else:
        info = native.device_info(device)
        cpus = (None if os.environ.get("REINFOCUS_BENCH_NO_PIN") == "1"
                else native.pin_to_numa_node(info["numa_node"], whole_process=True))
    try:
        now = sorted(os.sched_getaffinity(0))
    except AttributeError:
        now = None
    return dict(info, rank=rank, host=socket.gethostname(), pinned=cpus is not None, cpus=compact_ranges(now))


def compact_ranges(values):
    """[0, 1, 2, 3, 8, 9] -> "0-3,8-9" (a rank's CPU set in the JSON line)."""
    if not values:
        return None
    runs, start, prev = [], values[0], values[0]
    for v in values[1:] + [None]:
        if v is None or v != prev + 1:
            runs.append(str(start) if start == prev else f"{start}-{prev}")
            start = v
        prev = v
    return ",".join(runs)


def check_distinct_devices(rows, allow_shared):
    """An N-GPU line has to come from N distinct GPUs (of one host): returns an error string or None."""
    seen = {}
    for row in rows:
        key = (row.get("host"), row["pci_bus_id"])
        if key in seen and not allow_shared:
            return (f"ranks {seen[key]} and {row['rank']} both ran on the GPU at {row['pci_bus_id']}: not an "
                    f"{len(rows)}-GPU measurement (REINFOCUS_BENCH_DEVICE allows it for rehearsals)")
        seen.setdefault(key, row["rank"])
    return None


def shard_plan(rank, envs_per_gpu, frame):
    """Rank r owns global envs [r*E, (r+1)*E) and therefore the RNG states a single-device
    run of all envs would use for them (pixel index = e*h*w + y*w + x, render.py:217)."""
    return {"first_env": rank * envs_per_gpu, "first_state_index": rank * envs_per_gpu * frame * frame}


# ---------------------------------------------------------------------------------------------
# CPU baseline (rank 0, N = 1 only)
# ---------------------------------------------------------------------------------------------
def host_cores():
    """Every core this process may run on."""
    try:
        return len(os.sched_getaffinity(0))
    except AttributeError:
        return os.cpu_count() or 1


def cgroup_cpu_quota():
    """CPUs' worth of time this process's cgroup may use (cpu.max), or None if unlimited / unknown."""
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        return None if quota == "max" else float(quota) / float(period)
    except (OSError, ValueError):
        return None


def cpu_pass(orc, frame, spp, n, cores, states, rng):
    """One render + focus pass of the C oracle over n synthetic environments; seconds."""
    from reinfocus_amd.graphics import camera, world

    targets = rng.uniform(5, 10, n).astype(np.float32)
    focus = rng.uniform(5, 10, n).astype(np.float32)
    cams = camera.FastCameras()
    cams.update(focus)
    worlds = world.FastWorlds()
    worlds.update(targets)
    dyn, origin, u, v, lens = cams.device_data()
    t0 = time.perf_counter()
    frames = orc.render(dyn, worlds.device_data(), frame, frame, spp, states,
                        cs=orc.cam_static(origin, u, v, float(lens)), n_threads=cores)
    orc.focus_values(frames, 15, cores)
    return time.perf_counter() - t0


def cpu_baseline(frame, spp, n_envs, device):
    """The CPU oracle (a port: the reference's numba-CUDASIM path cannot run here) timed on this
    host's cores (all of them offered; the team size that is fastest is used and stated) on a
    bounded sample of the same workload (~12 s of work), every environment with its own
    numba-seeded RNG states.  Seeding is untimed: the sample's states come from
    rf_seed (bit-identical to numba's sequential seeding, tests/test_gpu_parity.py), which
    takes milliseconds where the sequential definition takes ~0.5 s per million states.
    BASELINE.json configs[0] and configs[1] are small enough to be timed in full."""
    # (OMP_PROC_BIND is honoured when the caller sets it -- it has to be in the environment before the oracle, the process's
    # first OpenMP code, is loaded -- and recorded in the line; it is not set here: on the GPU boxes, whose jobs get a 16-CPU
    # share of a 128-thread host, "close" packs the team onto SMT siblings and costs 30 %: profiles/r06_ab.txt section 6)
    from oracle import oracle as orc
    from reinfocus_amd import _native

    visible = host_cores()
    rng = np.random.Generator(np.random.PCG64DXSM(0))
    ctx = _native.Context(device)

    def seeded(n, h):
        ctx.seed(n * h * h, 0, 0)
        return ctx.get_states()

    try:
        # All host cores are the baseline's to use, but a GPU box hands a job only its share of the
        # host's CPU time (one GPU's worth): more runnable threads than that share only add
        # contention.  The thread count is therefore chosen by measurement -- every visible core
        # and a few smaller teams on the same 64-env sample -- and all the rates are reported.
        sweep = {}
        sample = seeded(64, frame)
        for team in sorted({t for t in (8, 16, 32, 64, 128) if t < visible} | {visible}):
            sweep[team] = 64 / cpu_pass(orc, frame, spp, 64, team, sample.copy(), rng)
        cores = max(sweep, key=sweep.get)
        if n_envs <= 0:
            n_envs = int(max(cores, min(4096, round(8.0 * sweep[cores]))))
        # the same sample on both builds of the oracle (oracle/Makefile): -O2, the checker's, and -O3 -march=x86-64-v3 --
        # both with the parity flags (-ffp-contract=off -fno-fast-math) and both checked against the goldens; the faster one
        # is the baseline's value
        builds = []
        for name in ("o2", "o3"):
            flags = orc.use_build(name)
            seconds = cpu_pass(orc, frame, spp, n_envs, cores, seeded(n_envs, frame), rng)
            builds.append({"build": name, "compiler": flags, "value": n_envs / seconds, "seconds": seconds})
        best = max(builds, key=lambda b: b["value"])
        orc.use_build(best["build"])
        dt = best["seconds"]
        others = []
        for label, n, h, s in (("configs[0]: 1 env x 64x64 x 1 spp", 1, 64, 1),
                               ("configs[1]: 256 envs x 128x128 x 4 spp", 256, 128, 4)):
            team = min(cores, max(1, n))
            t = min(cpu_pass(orc, h, s, n, team, seeded(n, h), rng) for _ in range(3))
            others.append({"workload": label + " (full)", "value": n / t, "unit": "env-steps/s", "seconds": t,
                           "cores": team})
    finally:
        ctx.close()
        orc.use_build("o2")
    return {"value": n_envs / dt, "unit": "env-steps/s", "cores": cores, "host_cpu_count": os.cpu_count(),
            "builds": builds, "omp_proc_bind": os.environ.get("OMP_PROC_BIND"),
            "visible_cores": visible, "cgroup_cpu_quota": cgroup_cpu_quota(),
            "thread_sweep_env_steps_per_s": {str(k): v for k, v in sweep.items()},
            "kind": "port",
            "sample": f"{n_envs} envs x {frame}x{frame} x {spp} spp, each with its own seeded RNG states: one "
                      f"render + focus pass of the C oracle (the faster of its -O2 and -O3 -march=x86-64-v3 builds: "
                      f"{best['build']}; OpenMP, {cores} threads: the fastest team of a sweep "
                      f"up to all {visible} visible cores; OMP_PROC_BIND={os.environ.get('OMP_PROC_BIND')}), {dt:.1f} s",
            "other_configs": others}


# ---------------------------------------------------------------------------------------------
def main(argv=None):
    raw = sys.argv[1:] if argv is None else list(argv)
    args = parse_args(raw)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ and not args.sharded_env:
        raise SystemExit(launch_ranks(args.gpus, raw))

    if args.plumbing_test and os.environ.get("REINFOCUS_BENCH_HANG_RANK") == os.environ.get("RANK", "0"):
        import signal  # test hook: a rank that hangs and ignores SIGTERM must still be ended (by PID)

        signal.signal(signal.SIGTERM, signal.SIG_IGN)
        time.sleep(3600)
    if args.plumbing_test and os.environ.get("REINFOCUS_BENCH_FAIL_EARLY_RANK") == os.environ.get("RANK", "0"):
        sys.exit(3)  # test hook: a rank that dies before the rendezvous
    ranks = Ranks(args.gpus, single_process=args.sharded_env)
    if args.plumbing_test and os.environ.get("REINFOCUS_BENCH_FAIL_RANK") == str(ranks.rank):
        sys.exit(3)  # test hook: a rank that dies must fail the whole launch
    n_local, frame, spp = args.envs_per_gpu, args.frame, args.spp
    plan = shard_plan(ranks.rank, n_local, frame)
    n_here = n_local * (args.gpus if args.sharded_env else 1)  # environments this process steps

    env = None
    ctx = None
    if not args.plumbing_test:
        # the HIP library is loaded before anything else can pull a second HIP runtime in
        from reinfocus_amd import _native
        from reinfocus_amd.environments import harness

        # REINFOCUS_BENCH_DEVICE: every rank / every shard on that one device (rehearsals on a 1-GPU box)
        pinned = os.environ.get("REINFOCUS_BENCH_DEVICE")
        if _native.device_count() < (args.gpus if args.sharded_env and pinned is None else 1):
            raise SystemExit("bench.py needs a GPU per rank: reinfocus_amd has no CPU fallback")
        # a launcher that narrows every rank's view to its own GPU (HIP_VISIBLE_DEVICES per rank) leaves one visible
        # device, index 0, whatever LOCAL_RANK says; the table of PCI bus ids below is what proves the ranks' GPUs distinct
        visible = _native.device_count()
        device = int(pinned) if pinned is not None else (ranks.local_rank if ranks.local_rank < visible
                                                         else ranks.local_rank % visible)
        common = dict(num_envs=n_here, frame_height=frame, samples_per_pixel=spp, seed=ranks.rank)
        if args.sharded_env:
            devices = [int(pinned)] * args.gpus if pinned is not None else list(range(args.gpus))
            env = harness.ShardedVectorDiscreteSteps(devices=devices, numa_pin=os.environ.get("REINFOCUS_BENCH_NO_PIN") != "1",
                                                     **common)
            contexts = [shard.ctx for shard in env._shards]
            placements = [dict(p, rank=g, host=os.uname().nodename, pinned=p["cpus"] is not None,
                               cpus=compact_ranges(p["cpus"])) for g, p in enumerate(env.placements)]
        else:
            # which physical GPU, and this process on that GPU's NUMA node, before the context (its stream, its pinned
            # staging buffers, the environment's host arrays) exists
            placements = [rank_placement(ranks.rank, device, _native)]
            if args.env == "device":
                env = harness.DeviceVectorDiscreteSteps(device=device, first_state_index=plan["first_state_index"], **common)
                contexts = [env._ctx]
            else:
                env = harness.VectorDiscreteSteps(device=device, first_state_index=plan["first_state_index"],
                                                  host_frames=args.env == "literal", **common)
                contexts = [env._renderer._ctx]
                if args.env == "literal":  # host arrays are scored on vision's scratch context
                    from reinfocus_amd import vision

                    contexts.append(vision._scratch_context())
        ctx = contexts[0]
        env.reset()
        pixels_before_first_step = _native.pixels_rendered()
        kernel_before_first_step = ctx.render_kernel_name()
    else:
        placements = [rank_placement(ranks.rank, ranks.local_rank)]
    if not args.sharded_env:
        placements = [row for rows in ranks.gather(placements) for row in rows]
    shared_gpu = check_distinct_devices(placements, allow_shared=os.environ.get("REINFOCUS_BENCH_DEVICE") is not None)
    if shared_gpu:
        if env is not None:
            env.close()
        ranks.close()
        raise SystemExit("bench.py: " + shared_gpu)
    action_rng = np.random.Generator(np.random.PCG64DXSM(1000 + ranks.rank))

    gathered_bytes = [0]
    gather_seconds = [0.0]  # (rank 0's time inside the gather: waiting for the slowest rank included)

    def gather_step(*arrays):
        t_gather = time.perf_counter()
        gathered_bytes[0] = ranks.gather_step(*arrays)
        gather_seconds[0] += time.perf_counter() - t_gather

    def one_step():
        actions = action_rng.integers(0, 13, n_here)
        if env is None:  # (plumbing: a fake block goes through the same gather)
            time.sleep(0.002)
            gather_step(np.zeros((n_here, 4), np.float32), np.zeros(n_here), np.zeros(n_here, bool), np.ones(n_here, bool))
            return 0
        obs, rewards, term, trunc, _ = env.step(actions)
        # one process per GPU: the ranks' blocks are gathered on rank 0, as the one vector environment they stand for would
        # return them (the sharded product object concatenates inside step() itself)
        gather_step(obs, rewards, term, trunc)
        return int((term | trunc).sum())

    for _ in range(args.warmup):
        one_step()

    def on_contexts(call):
        if args.sharded_env:
            env._each(lambda shard: call(shard.ctx))
        else:
            for c in contexts:
                call(c)

    if env is not None:
        if not args.no_kernel_timing:
            on_contexts(lambda c: c.timing(True))
        on_contexts(lambda c: c.synchronize())
    ranks.barrier()
    gather_seconds[0] = 0.0
    t0 = time.perf_counter()
    resets = 0
    for _ in range(args.steps):
        resets += one_step()
    if env is not None:
        on_contexts(lambda c: c.synchronize())
    elapsed_local = time.perf_counter() - t0
    ranks.barrier()
    elapsed = ranks.reduce(elapsed_local, "MAX")
    total_resets = ranks.reduce(resets, "SUM")
    # every rank's own time per step (its steps + its share of the gather), in rank order: a straggling rank -- a GPU behind
    # another PCIe root, a rank on the wrong NUMA node -- is visible in the line, not only in the maximum
    per_rank_ms = [1000.0 * seconds / max(args.steps, 1) for seconds in ranks.gather(elapsed_local)]

    timing = None
    if env is not None and not args.no_kernel_timing:
        if args.sharded_env:
            timing = env._each(lambda shard: shard.ctx.timing_read())[0]
        else:  # (the literal route renders on one context and scores on another)
            reads = [c.timing_read() for c in contexts]
            timing = {key: sum(r[key] for r in reads) for key in reads[0]}
    if timing is not None and args.sharded_env:  # (a shard's context is only ever touched from its own thread)
        all_kernel_ms = sum(sum(t[k] for k in ("render_ms", "focus_ms"))
                            for t in env._each(lambda shard: shard.ctx.timing_read()))
    else:
        all_kernel_ms = sum(timing[k] for k in ("render_ms", "focus_ms")) if timing is not None else None
    n_gpus = args.gpus if args.sharded_env else ranks.world
    total_envs = n_local * n_gpus
    value = total_envs * args.steps / elapsed if env is not None else None

    out = None
    if ranks.rank == 0:
        out = {
            "metric": "vectorised env steps/sec @ 4096 envs 256x256x16spp; 1->8 GPU scaling",
            "value": value,
            "unit": "env-steps/s",
            "n_gpus": n_gpus,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": 1000.0 * elapsed / args.steps,
            "per_rank_ms_per_step": per_rank_ms,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic" if env is not None else "none (plumbing test, not a measurement)",
            "config": {
                "workload": f"DiscreteSteps-v0 vector env, {n_local} envs/GPU x {frame}x{frame} x {spp} spp "
                            f"({baseline_config_name(n_local, frame, spp, n_gpus)})",
                "envs_per_gpu": n_local,
                "total_envs": total_envs,
                "frame": frame,
                "spp": spp,
                "auto_resets_per_step": total_resets / max(args.steps, 1),
                "sharding": ("one process, one rf_ctx + host thread per GPU (ShardedVectorDiscreteSteps): step() returns the "
                             "concatenated arrays" if args.sharded_env else
                             "one process per GPU, contiguous env ranges; every step's observations / rewards / flags gathered "
                             "on rank 0 over gloo inside the timed region") +
                            ", no data-path collective",
                # what rank 0 holds after a step: N x 29 B (value = all ranks' environments / the time to step AND gather them)
                "host_gather_bytes_per_step": gathered_bytes[0],
                # rank 0's share of a step spent in that gather (the wait for the slowest rank's step included)
                "host_gather_ms_per_step": 1000.0 * gather_seconds[0] / max(args.steps, 1),
                "env_glue": {"device": "device-resident (rf_env_step)", "host": "host numpy (harness), frames stay in HBM",
                             "literal": "host numpy (harness) over the literal stub: rf_render(host_out) + "
                                        "rf_upload_frames + rf_focus, frames cross PCIe twice per render"}[args.env],
            },
            # one row per rank (per shard with --sharded-env): an N-GPU value comes from N distinct GPUs
            "devices": placements,
        }
        if args.env == "literal" and env is not None:
            per_step = n_local * frame * frame * 3 * (1.0 + total_resets / max(args.steps, 1) / max(total_envs, 1))
            out["pcie_bytes_per_step_per_gpu"] = {"device_to_host": per_step, "host_to_device": per_step}
        if env is not None:
            # the renders before the first step (reset, the 13-environment extrema render of cached_focus_extrema:
            # same frame size and sample count) use the single-pass instance of the kernel, the fused steps the
            # two-pass one (separate launches: the same instance throughout)
            by_kernel = {kernel_key(kernel_before_first_step): pixels_before_first_step}
            stepped = kernel_key(ctx.render_kernel_name())
            by_kernel[stepped] = by_kernel.get(stepped, 0) + _native.pixels_rendered() - pixels_before_first_step
            out["render_pixels_by_kernel"] = by_kernel
            out["render_pixels_before_first_step"] = pixels_before_first_step
        if timing is not None:
            # full renders + the partial auto-reset renders of rank 0 (of shard 0 with --sharded-env)
            resets_here = resets if not args.sharded_env else resets / n_gpus
            pixels = (args.steps * n_local + resets_here) * frame * frame
            render_s = timing["render_ms"] / 1000.0
            focus_s = timing["focus_ms"] / 1000.0
            achieved = RENDER_BYTES_PER_PIXEL * pixels / render_s / 1e9
            launches = max(timing["render_launches"], 1)
            kernel = ctx.render_kernel_name()
            profile, why_not, pmc_failure = None, None, None
            if not args.no_pmc and n_gpus == 1 and not args.sharded_env:
                profile, pmc_failure = measure_pmc(args, kernel)
            if profile is None:
                profile, why_not = committed_profile(kernel, frame, spp)
            traffic, valu = roofline_from_profile(profile, pixels / launches) if profile else (None, None)
            live = profile is not None and profile["file"] is None
            out["roofline"] = {
                "bound": "hbm",
                "kernel": kernel,
                "achieved": achieved,
                "peak": HBM_PEAK_GBPS,
                "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBPS,
                "measured_in_this_run": ["achieved", "frac", "avg_launch_ms", "launches", "samples_per_s"],
                "algorithmic_bytes_per_launch": RENDER_BYTES_PER_PIXEL * pixels / launches,
                "algorithmic_bytes_per_pixel": RENDER_BYTES_PER_PIXEL,
                "avg_launch_ms": timing["render_ms"] / launches,
                "launches": timing["render_launches"],
                "samples_per_s": pixels * spp / render_s,
                "traffic": traffic,
                "traffic_unit": "HBM bytes per launch (PMC FETCH_SIZE x2 + WRITE_SIZE)",
                "traffic_from_committed_profile": traffic is not None and not live,
                "traffic_measured_by": ("rocprofv3 --pmc child runs of this invocation" if live else
                                        "committed profile" if traffic is not None else None),
                "traffic_profile": ({k: profile[k] for k in ("file", "commit", "config")} if profile
                                    else {"unavailable": why_not}),
                "note": "HBM is the nominated roof; the kernel is bound by VALU issue (64-bit xoroshiro128+ "
                        "draws and rejection loops fixed by parity): see roofline_valu and DESIGN.md",
            }
            out["roofline_valu"] = valu if valu is not None else {"unavailable": why_not, "from_committed_profile": True}
            if pmc_failure:
                out["roofline"]["pmc_failure"] = pmc_failure
            focus_gbps = FOCUS_BYTES_PER_PIXEL * pixels / focus_s / 1e9 if focus_s > 0 else None
            out["focus_kernel"] = {  # north_star's second kernel: its nominated roof (HBM) is its real one
                "kernel": "focus_kernel_roll" if frame % 4 == 0 and frame >= 8 else "focus_kernel",
                "bound": "hbm",
                "achieved_GBps": focus_gbps,
                "peak_GBps": 8000.0,
                "frac": focus_gbps / 8000.0 if focus_gbps else None,
                "algorithmic_bytes_per_pixel": FOCUS_BYTES_PER_PIXEL,
                "avg_launch_ms": timing["focus_ms"] / max(timing["focus_launches"], 1),
                "launches": timing["focus_launches"],
            }
            out["kernel_time_frac_of_wall"] = (render_s + focus_s) / elapsed_local
            if args.sharded_env:
                # (per-context kernel times overlap when several contexts share a device: the sum of the
                # event spans says nothing about the host side; compare wall_ms_per_step with n_gpus x the
                # one-context step instead)
                out["sharded_env"] = {"contexts": len(contexts), "devices": env.devices,
                                      "wall_ms_per_step": 1000.0 * elapsed_local / args.steps,
                                      "sum_of_kernel_event_spans_ms_per_step": all_kernel_ms / args.steps}
        if env is not None and not args.no_cpu_baseline and n_gpus == 1:
            out["cpu_baseline"] = cpu_baseline(frame, spp, args.cpu_baseline_envs, ctx.device)
    if env is not None:
        env.close()
    ranks.close()
    if out is not None:
        print(json.dumps(out), flush=True)
    return out


if __name__ == "__main__":
    main()
