"""Development tool: table of tools/bench_general.py results per library from a log of tools/ab_general.sh-style runs.
usage: python tools/tab_general.py gpurun_out/<dir>/timing.txt"""
import collections
import re
import sys

res = collections.defaultdict(lambda: collections.defaultdict(list))
lib = None
for line in open(sys.argv[1]):
    if line.startswith("=="):
        lib = line.split()[1].split("/")[-1]
        continue
    m = re.match(r"(\w+): .* ([\d.]+) G samples/s", line)
    if m:
        res[lib][m.group(1)].append(float(m.group(2)))
scenes = ["one_rect", "one_sphere", "two_sphere", "mixed"]
print(f"{'lib':28s}" + "".join(f"{s:>18s}" for s in scenes))
for lib, v in res.items():
    print(f"{lib:28s}" + "".join(f"{'/'.join(f'{x:.1f}' for x in v[s]):>18s}" for s in scenes))
