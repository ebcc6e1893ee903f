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
    spans = collections.defaultdict(float)  # ns of the profiled launches, from the pass that holds GRBM_GUI_ACTIVE
    for d in ("pmc_sq", "pmc_sq2", "pmc_mix", "pmc_mix2", "pmc_fetch", "pmc_write"):
        for f in sorted(glob.glob(os.path.join(src, d, "*", "*_counter_collection.csv")), key=os.path.getmtime)[-1:]:
            seen = set()
            for r in csv.DictReader(open(f)):
                k = r["Kernel_Name"].split("(")[0]
                if r["Counter_Name"] == "SQ_INSTS_VALU" and d != "pmc_sq":
                    totals[k]["SQ_INSTS_VALU@" + d] += float(r["Counter_Value"])  # the denominator of that pass
                    continue
                totals[k][r["Counter_Name"]] += float(r["Counter_Value"])
                if r["Dispatch_Id"] not in seen:
                    seen.add(r["Dispatch_Id"])
                    if d == "pmc_sq":
                        launches[k] += 1
                    if d == "pmc_sq2" and "Start_Timestamp" in r:
                        spans[k] += float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
    # pixels the profiled process really rendered (bench.py prints rf_pixels_rendered per kernel)
    pixels_by_kernel = {}
    for log in ("pmc_sq.log",):
        path = os.path.join(src, log)
        if os.path.exists(path):
            for line in open(path):
                if line.startswith("{") and "render_pixels_by_kernel" in line:
                    pixels_by_kernel = json.loads(line)["render_pixels_by_kernel"]
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
        key = k.replace("void ", "").replace("rf::", "").replace(" ", "")
        if key.startswith("render_kernel_coop2<") and key.endswith(",false>"):  # (bench.kernel_key: the library's own name)
            key = key[:-len(",false>")] + ">"
        if key in pixels_by_kernel and e.get("SQ_WAVES"):
            e["pixels"] = float(pixels_by_kernel[key])
            e["pixels_per_wave"] = e["pixels"] / e["SQ_WAVES"]
            e["pixels_from"] = "rf_pixels_rendered of the profiled run"
        elif "render_kernel" in k:
            e["pixels_per_wave"] = 192 if ("render_kernel_coop2" in k or "render_kernel_wave" in k) else 64  # (the wave kernel: K = 3)
            e["pixels_from"] = "SQ_WAVES x pixels per wave (padded lanes included)"
        # sustained shader clock under this kernel: GRBM_GUI_ACTIVE counts every XCD's busy cycles
        if e.get("GRBM_GUI_ACTIVE") and spans.get(k):
            e["sustained_clock_ghz"] = e["GRBM_GUI_ACTIVE"] / 8.0 / spans[k]
        # dynamic instruction mix (counters of the pmc_mix passes, as shares of that pass's SQ_INSTS_VALU)
        mix_total = e.get("SQ_INSTS_VALU@pmc_mix")
        if mix_total:
            e["valu_mix"] = {name[len("SQ_INSTS_VALU_"):].lower(): e[name] / mix_total
                             for name in sorted(e) if name.startswith("SQ_INSTS_VALU_") and "@" not in name
                             and name not in ("SQ_INSTS_VALU_FMA_F64", "SQ_INSTS_VALU_ADD_F64", "SQ_INSTS_VALU_MUL_F64")}
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
    # static VALU mix of the profiled render kernel instance by gfx950 issue class (tools/isa_mix.py on the
    # listing `make -C reinfocus_amd/csrc asm` writes), with the per-class costs measured by tools/ubench/pairbench
    kernel = (meta["config"] or {}).get("kernel") or ""
    match = __import__("re").match(r"render_kernel_coop2<(true|false), (\d+), (\d+), (\d+)(, true|, false)?>", kernel)
    listing = os.path.join(root, "..", "reinfocus_amd", "csrc", "rf_abi_render.gfx950.s")  # (the render kernels' unit)
    strip = __import__("re").match(r"render_kernel_coop2_strip<(\d+), (\d+)>", kernel)
    if (match or strip) and os.path.exists(listing):
        sys.path.insert(0, os.path.join(root, "..", "tools"))
        import isa_mix

        if strip:  # (both tile shapes' code)
            mangled = "_ZN2rf25render_kernel_coop2_stripILi%sELi%sEEEvNS_10RenderArgsE" % (strip.group(1), strip.group(2))
        else:
            mangled = "_ZN2rf19render_kernel_coop2ILb%dELi%sELi%sELi%sELb%dEEEvNS_10RenderArgsE" % (
                1 if match.group(1) == "true" else 0, match.group(2), match.group(3), match.group(4),
                1 if match.group(5) == ", true" else 0)
        counts = isa_mix.mix(isa_mix.kernel_lines(listing, mangled))
        total = float(sum(counts.values()))
        if total:
            shares = {c: n / total for c, n in counts.items()}
            no_overlap = sum(shares[c] * isa_mix.COST[c] for c in shares)
            fp_hidden = sum(shares[c] * isa_mix.COST[c] for c in shares if c != "fast-fp")
            meta["static_valu_mix"] = {
                "shares": shares, "cycles_per_class": isa_mix.COST,
                "cycles_per_instruction_if_nothing_overlaps": no_overlap,
                "cycles_per_instruction_if_fast_fp_is_hidden": fp_hidden,
                "attainable_insts_per_cycle_per_simd": [1.0 / no_overlap, 1.0 / fp_hidden],
                "source": "static mix of the kernel's ISA (tools/isa_mix.py); class costs: tools/ubench/pairbench, "
                          "profiles/r03_pairbench_8w.txt",
            }
    out["_meta"] = meta
    with open(os.path.join(root, tag + "_pmc.json"), "w") as f:
        json.dump(out, f, indent=1, sort_keys=True)
    print(open(os.path.join(root, tag + "_kernel_stats.csv")).read())
    print(json.dumps({k: {m: v[m] for m in v if m.startswith(("hbm", "valu", "launches", "wait", "inst_wait", "salu", "commit", "config", "sustained", "pixels"))}
                      for k, v in out.items()}, indent=1))


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else "r01")
