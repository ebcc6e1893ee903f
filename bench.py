"""Headline benchmark: vectorised env-steps/s of DiscreteSteps-v0 on MI355X.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

A "step" is one VectorDiscreteSteps.step(actions) over this rank's environments: host
transform / ender / reward (numpy), scene upload, the render kernel, the focus kernel, an
8 B/env D2H and the same-step partial render of the environments that just ended
(SURVEY.md section 8(d)).  Workload = BASELINE.json configs[2]: 4096 envs x 256x256 x 16 spp
per GPU; with N > 1 every rank owns 4096 more envs (weak scaling, BASELINE configs[3]) and
its own slice of the RNG state sequence.  There is no data-path collective: environments are
independent, ranks only meet at the timing barriers (gloo, host side).

Prints ONE JSON line (rank 0).  value = all ranks' env-steps / max-over-ranks seconds, with
inputs resident in HBM (the only per-step host traffic is 44 B/env of scene parameters in
and 8 B/env of focus values out, both inside the timed region).
"""

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
RENDER_BYTES_PER_PIXEL = 35  # 16 B state read + 16 B state write + 3 B frame write
FOCUS_BYTES_PER_PIXEL = 3    # frame read


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--envs-per-gpu", type=int, default=4096)
    ap.add_argument("--frame", type=int, default=256)
    ap.add_argument("--spp", type=int, default=16)
    ap.add_argument("--cpu-baseline-envs", type=int, default=0, help="0 = choose ~15 s of CPU work")
    ap.add_argument("--env", choices=["device", "host"], default="device",
                    help="device: whole step resident on the GPU (rf_env_*); host: numpy glue around "
                         "rf_render / rf_focus (identical results, tests/test_gpu_environment.py)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-timing", action="store_true",
                    help="no per-kernel HIP events (and no roofline object): lets small configurations replay "
                         "their step as one hipGraph")
    ap.add_argument("--plumbing-test", action="store_true",
                    help="no GPU work: exercises only the multi-rank plumbing (CPU tests)")
    return ap.parse_args(argv)


class Ranks:
    """Host-side rendezvous of the one-process-per-GPU ranks (gloo; no GPU tensors)."""

    def __init__(self, gpus):
        self.rank = int(os.environ.get("RANK", "0"))
        self.local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        if self.world != gpus:
            raise SystemExit(f"--gpus {gpus} but WORLD_SIZE={self.world}: launch with torch.distributed.run "
                             f"--nproc-per-node {gpus}")
        self.dist = None
        if self.world > 1:
            import torch.distributed as dist

            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            dist.init_process_group(backend="gloo", rank=self.rank, world_size=self.world)
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

    def close(self):
        if self.dist is not None:
            self.dist.destroy_process_group()


def pmc_bytes_per_pixel():
    """HBM bytes per rendered pixel of the render kernel from the committed rocprofv3 PMC
    passes (profiles/<tag>_pmc.json: FETCH_SIZE doubled for gfx950's wide-read under-count,
    WRITE_SIZE as is, divided by SQ_WAVES * pixels per wave; partial tiles at the frame's edge count
    as whole ones: 258 instead of 256 rows at the headline size).  None if no profile is present."""
    import glob

    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "*_pmc.json")))
    if not files:
        return None, None
    data = json.load(open(files[-1]))
    for name, e in data.items():
        if "render_kernel" in name and e.get("SQ_WAVES") and "hbm_read_bytes_corrected" in e:
            pixels = e["SQ_WAVES"] * float(e.get("pixels_per_wave", 64))
            return (e["hbm_read_bytes_corrected"] + e["hbm_write_bytes"]) / pixels, os.path.basename(files[-1])
    return None, None


def pmc_valu_summary():
    """VALU-side figures of the render kernel from the same PMC summary (the kernel is bound by
    vector-ALU issue, not by HBM): instructions per wave and per 64 pixels, lane utilisation, SIMD
    cycles available per VALU instruction issued."""
    import glob

    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "*_pmc.json")))
    if not files:
        return None
    for name, e in json.load(open(files[-1])).items():
        if "render_kernel" in name and e.get("valu_insts_per_wave"):
            per_wave = float(e.get("pixels_per_wave", 64))
            return {"insts_per_wave": e["valu_insts_per_wave"],
                    "insts_per_64_pixels": e["valu_insts_per_wave"] * 64.0 / per_wave,
                    "lane_utilisation": e.get("valu_lane_utilisation"),
                    "simd_cycles_per_inst": e.get("simd_cycles_per_valu_inst"),
                    "source": os.path.basename(files[-1])}
    return None


def shard_plan(rank, envs_per_gpu, frame):
    """Rank r owns global envs [r*E, (r+1)*E) and therefore the RNG states a single-device
    run of all envs would use for them (pixel index = e*h*w + y*w + x, render.py:217)."""
    return {"first_env": rank * envs_per_gpu, "first_state_index": rank * envs_per_gpu * frame * frame}


def cpu_baseline(frame, spp, n_envs):
    """The CPU oracle (a port: the reference's numba-CUDASIM path is not runnable here)
    timed on this host's cores on a bounded sample of the same workload (~12 s of work).
    Seeding is untimed; to keep it short the sample's RNG states are the first env's
    numba-seeded states tiled over the sampled envs (same arithmetic per pixel)."""
    from oracle import oracle as orc
    from reinfocus_amd.graphics import camera, world

    cores = min(os.cpu_count() or 1, 16)
    rng = np.random.Generator(np.random.PCG64DXSM(0))
    one_env = orc.seed_states(frame * frame, 0)

    def run(n):
        targets = rng.uniform(5, 10, n).astype(np.float32)
        focus = rng.uniform(5, 10, n).astype(np.float32)
        cams = camera.FastCameras()
        cams.update(focus)
        worlds = world.FastWorlds()
        worlds.update(targets)
        dyn, origin, u, v, lens = cams.device_data()
        states = np.ascontiguousarray(np.tile(one_env, (n, 1)))
        t0 = time.perf_counter()
        frames = orc.render(dyn, worlds.device_data(), frame, frame, spp, states,
                            cs=orc.cam_static(origin, u, v, float(lens)), n_threads=cores)
        orc.focus_values(frames, 15, cores)
        return time.perf_counter() - t0

    if n_envs <= 0:
        dt = run(cores)  # calibration: one env per thread
        n_envs = int(max(cores, min(4096, round(12.0 * cores / max(dt, 1e-3)))))
    dt = run(n_envs)
    return {"value": n_envs / dt, "unit": "env-steps/s", "cores": cores, "kind": "port",
            "sample": f"{n_envs} envs x {frame}x{frame} x {spp} spp, one render+focus pass of the C oracle "
                      f"(OpenMP, {cores} threads), {dt:.1f} s"}


def main(argv=None):
    args = parse_args(argv)
    ranks = Ranks(args.gpus)
    n_local, frame, spp = args.envs_per_gpu, args.frame, args.spp
    plan = shard_plan(ranks.rank, n_local, frame)

    env = None
    if not args.plumbing_test:
        # the HIP library is loaded before anything else can pull a second HIP runtime in
        from reinfocus_amd import _native
        from reinfocus_amd.environments import harness

        if _native.device_count() < 1:
            raise SystemExit("bench.py needs a GPU: reinfocus_amd has no CPU fallback")
        device = int(os.environ.get("REINFOCUS_BENCH_DEVICE", ranks.local_rank))
        env_cls = harness.DeviceVectorDiscreteSteps if args.env == "device" else harness.VectorDiscreteSteps
        env = env_cls(num_envs=n_local, frame_height=frame, samples_per_pixel=spp, seed=ranks.rank,
                      device=device, first_state_index=plan["first_state_index"])
        ctx = env._ctx if args.env == "device" else env._renderer._ctx
        env.reset()
    action_rng = np.random.Generator(np.random.PCG64DXSM(1000 + ranks.rank))

    def one_step():
        actions = action_rng.integers(0, 13, n_local)
        if env is None:
            time.sleep(0.002)
            return 0
        _, _, term, trunc, _ = env.step(actions)
        return int((term | trunc).sum())

    for _ in range(args.warmup):
        one_step()

    if env is not None:
        if not args.no_kernel_timing:
            ctx.timing(True)
        ctx.synchronize()
    ranks.barrier()
    t0 = time.perf_counter()
    resets = 0
    for _ in range(args.steps):
        resets += one_step()
    if env is not None:
        ctx.synchronize()
    elapsed_local = time.perf_counter() - t0
    ranks.barrier()
    elapsed = ranks.reduce(elapsed_local, "MAX")
    total_resets = ranks.reduce(resets, "SUM")

    timing = ctx.timing_read() if env is not None and not args.no_kernel_timing else None
    total_envs = n_local * ranks.world
    value = total_envs * args.steps / elapsed if env is not None else None

    out = None
    if ranks.rank == 0:
        out = {
            "metric": "vectorised env steps/sec @ 4096 envs 256x256x16spp; 1->8 GPU scaling",
            "value": value,
            "unit": "env-steps/s",
            "n_gpus": ranks.world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": 1000.0 * elapsed / args.steps,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic" if env is not None else "none (plumbing test, not a measurement)",
            "config": {
                "workload": f"DiscreteSteps-v0 vector env, {n_local} envs/GPU x {frame}x{frame} x {spp} spp "
                            f"(BASELINE.json configs[2]; weak-scaled per GPU = configs[3])",
                "envs_per_gpu": n_local,
                "total_envs": total_envs,
                "frame": frame,
                "spp": spp,
                "auto_resets_per_step": total_resets / max(args.steps, 1),
                "sharding": "independent env ranges per rank, no data-path collective",
                "env_glue": "device-resident (rf_env_step)" if args.env == "device" else "host numpy (harness)",
            },
        }
        if timing is not None:
            # full renders + the partial auto-reset renders of rank 0
            pixels = (args.steps * n_local + resets) * frame * frame
            render_s = timing["render_ms"] / 1000.0
            focus_s = timing["focus_ms"] / 1000.0
            achieved = RENDER_BYTES_PER_PIXEL * pixels / render_s / 1e9
            launches = max(timing["render_launches"], 1)
            pmc_bpp, pmc_file = pmc_bytes_per_pixel()
            out["roofline"] = {
                "bound": "hbm",
                "kernel": "render_kernel_coop2<POW2>",
                "achieved": achieved,
                "peak": HBM_PEAK_GBPS,
                "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBPS,
                "traffic": None if pmc_bpp is None else pmc_bpp * pixels / launches,
                "traffic_unit": "HBM bytes per launch (PMC FETCH_SIZE x2 + WRITE_SIZE)",
                "traffic_source": pmc_file,
                "algorithmic_bytes_per_launch": RENDER_BYTES_PER_PIXEL * pixels / launches,
                "algorithmic_bytes_per_pixel": RENDER_BYTES_PER_PIXEL,
                "avg_launch_ms": timing["render_ms"] / max(timing["render_launches"], 1),
                "launches": timing["render_launches"],
                "samples_per_s": pixels * spp / render_s,
                "valu": pmc_valu_summary(),
                "note": "VALU-bound by construction (64-bit xoroshiro128+ draws and rejection loops fixed "
                        "by parity): see DESIGN.md section Roofline",
            }
            out["focus_kernel"] = {
                "achieved_GBps": FOCUS_BYTES_PER_PIXEL * pixels / focus_s / 1e9,
                "avg_launch_ms": timing["focus_ms"] / max(timing["focus_launches"], 1),
                "launches": timing["focus_launches"],
            }
            out["kernel_time_frac_of_wall"] = (render_s + focus_s) / elapsed_local
        if env is not None and not args.no_cpu_baseline and ranks.world == 1:
            out["cpu_baseline"] = cpu_baseline(frame, spp, args.cpu_baseline_envs)
    if env is not None:
        env.close()
    ranks.close()
    if out is not None:
        print(json.dumps(out), flush=True)
    return out


if __name__ == "__main__":
    main()
