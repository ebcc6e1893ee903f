"""Condenses a gpurun_out/prof_<tag>/ directory (written by profiles/run_profiles.sh on the
GPU box) into the small files committed under profiles/: the rocprofv3 --stats kernel table
and the per-kernel PMC totals, with HBM bytes corrected as the MI355X guide prescribes
(FETCH_SIZE and WRITE_SIZE are in KiB; gfx950 FETCH_SIZE counts 128-B requests as 64 B for
wide coalesced reads, so the read side is doubled)."""

import collections
import csv
import glob
import json
import os
import shutil
import sys


def git_commit(root):
    import subprocess

    try:
        head = subprocess.check_output(["git", "-C", root, "rev-parse", "--short=12", "HEAD"], text=True).strip()
        dirty = subprocess.check_output(["git", "-C", root, "status", "--porcelain", "--", "reinfocus_amd", "bench.py"],
                                        text=True).strip()
        return head + ("+uncommitted" if dirty else "")
    except Exception:  # not a checkout
        return None


def main(tag):
    root = os.path.dirname(os.path.abspath(__file__))
    src = os.path.join(root, "..", "gpurun_out", "prof_" + tag)
    # gpurun merges successive calls into the same directory: keep the newest run of each pass
    stats = sorted(glob.glob(os.path.join(src, "trace", "*", "*_kernel_stats.csv")), key=os.path.getmtime)
    assert stats, "no kernel_stats.csv under " + src
    shutil.copy(stats[-1], os.path.join(root, tag + "_kernel_stats.csv"))
    totals = collections.defaultdict(lambda: collections.defaultdict(float))
    launches = collections.Counter()
    for d in ("pmc_sq", "pmc_sq2", "pmc_fetch", "pmc_write"):
        for f in sorted(glob.glob(os.path.join(src, d, "*", "*_counter_collection.csv")), key=os.path.getmtime)[-1:]:
            seen = set()
            for r in csv.DictReader(open(f)):
                k = r["Kernel_Name"].split("(")[0]
                totals[k][r["Counter_Name"]] += float(r["Counter_Value"])
                if d == "pmc_sq" and r["Dispatch_Id"] not in seen:
                    seen.add(r["Dispatch_Id"])
                    launches[k] += 1
    out = {}
    for k, v in totals.items():
        if k.startswith("__amd"):
            continue
        e = dict(v)
        e["launches"] = launches[k]
        if "FETCH_SIZE" in e:
            e["hbm_read_bytes_corrected"] = e["FETCH_SIZE"] * 1024 * 2
        if "WRITE_SIZE" in e:
            e["hbm_write_bytes"] = e["WRITE_SIZE"] * 1024
        # pixels a wave of the render kernels owns (render_kernel_coop2: 3 pixel sets per thread)
        if "render_kernel" in k:
            e["pixels_per_wave"] = 192 if "render_kernel_coop2" in k else 64
        if "SQ_INSTS_VALU" in e and e.get("SQ_WAVES"):
            e["valu_insts_per_wave"] = e["SQ_INSTS_VALU"] / e["SQ_WAVES"]
        # SIMD cycles the kernel had per VALU instruction it issued (GRBM_GUI_ACTIVE counts every XCD's
        # cycles; 1024 SIMDs): ~3.3 means the vector ALUs are saturated by this instruction mix
        if e.get("GRBM_GUI_ACTIVE") and e.get("SQ_INSTS_VALU"):
            e["simd_cycles_per_valu_inst"] = e["GRBM_GUI_ACTIVE"] / 8.0 * 1024.0 / e["SQ_INSTS_VALU"]
        if "SQ_THREAD_CYCLES_VALU" in e and e.get("SQ_ACTIVE_INST_VALU"):
            e["valu_lane_utilisation"] = e["SQ_THREAD_CYCLES_VALU"] / (e["SQ_ACTIVE_INST_VALU"] * 64)
        # VALU issue rate against its peak of one wave64 instruction per 2 cycles per SIMD
        if e.get("GRBM_GUI_ACTIVE") and e.get("SQ_INSTS_VALU"):
            e["valu_insts_per_cycle_per_simd"] = e["SQ_INSTS_VALU"] / (e["GRBM_GUI_ACTIVE"] / 8.0 * 1024.0)
            e["valu_issue_frac_of_peak"] = e["valu_insts_per_cycle_per_simd"] / 0.5
        if e.get("SQ_WAVE_CYCLES"):
            if "SQ_WAIT_ANY" in e:
                e["wait_share"] = e["SQ_WAIT_ANY"] / e["SQ_WAVE_CYCLES"]
            if "SQ_WAIT_INST_ANY" in e:
                e["inst_wait_share"] = e["SQ_WAIT_INST_ANY"] / e["SQ_WAVE_CYCLES"]
        if e.get("SQ_INSTS_VALU") and "SQ_INSTS_SALU" in e:
            e["salu_per_valu"] = e["SQ_INSTS_SALU"] / e["SQ_INSTS_VALU"]
        out[k] = e
    # what was profiled: the bench line of the PMC passes' command (run_profiles.sh writes it) and the commit
    meta = {"commit": git_commit(os.path.join(root, "..")), "config": None}
    bench_line = os.path.join(src, "bench.json")
    if os.path.exists(bench_line):
        for line in open(bench_line):
            if line.startswith("{"):
                b = json.loads(line)
                meta["config"] = {"envs": b["config"]["envs_per_gpu"], "frame": b["config"]["frame"],
                                  "spp": b["config"]["spp"], "kernel": b.get("roofline", {}).get("kernel")}
                shutil.copy(bench_line, os.path.join(root, tag + "_bench.json"))
    out["_meta"] = meta
    with open(os.path.join(root, tag + "_pmc.json"), "w") as f:
        json.dump(out, f, indent=1, sort_keys=True)
    print(open(os.path.join(root, tag + "_kernel_stats.csv")).read())
    print(json.dumps({k: {m: v[m] for m in v if m.startswith(("hbm", "valu", "launches", "wait", "inst_wait", "salu", "commit", "config"))}
                      for k, v in out.items()}, indent=1))


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else "r01")
