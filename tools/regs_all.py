"""Development tool: compact table of the registers / scratch / LDS / occupancy of every kernel in
libreinfocus_hip.so's source (`make asm` with -Rpass-analysis=kernel-resource-usage).
usage: python tools/regs_all.py [EXTRA flags ...]    e.g.  python tools/regs_all.py -DRF_FOO=1"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    extra = " ".join(sys.argv[1:])
    out = subprocess.run(["make", "-B", "-C", os.path.join(ROOT, "reinfocus_amd", "csrc"), "asm", f"EXTRA={extra}"],
                         capture_output=True, text=True)
    if out.returncode != 0:
        sys.stderr.write(out.stderr[-4000:])
        sys.exit(out.returncode)
    rows, cur = [], None
    for line in out.stderr.splitlines():
        m = re.search(r"remark:\s+(Function Name|VGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|SGPRs Spill|VGPRs Spill|LDS Size \[bytes/block\]|SGPRs): (\S+)", line)
        if not m:
            continue
        key, val = m.group(1), m.group(2)
        if key == "Function Name":
            cur = {"name": subprocess.run(["c++filt", val], capture_output=True, text=True).stdout.strip()}
            rows.append(cur)
        elif cur is not None:
            cur[key.split(" [")[0]] = val
    print(f"{'kernel':70s} vgpr sgpr scratch occ sspill vspill   lds")
    for r in rows:
        name = re.sub(r"\(.*", "", r["name"]).replace("rf::", "").replace("void ", "")
        g = lambda k: str(r.get(k, "-"))
        print(f"{name:70s} {g('VGPRs'):>4s} {g('SGPRs'):>4s} {g('ScratchSize'):>7s} {g('Occupancy'):>3s} "
              f"{g('SGPRs Spill'):>6s} {g('VGPRs Spill'):>6s} {g('LDS Size'):>5s}")


if __name__ == "__main__":
    main()
