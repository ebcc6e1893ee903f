#!/bin/bash
# Development tool: HBM traffic of the render kernel for the current build (two PMC passes).
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/traffic
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
BENCH="python3 $ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- $BENCH > $OUT/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write -- $BENCH > $OUT/write.log 2>&1
python3 - <<PY
import csv,glob
def tot(d,c):
    s=0;n=0
    for f in glob.glob('$OUT/'+d+'/**/*counter_collection.csv',recursive=True):
        for r in csv.DictReader(open(f)):
            if 'render_kernel' in r['Kernel_Name'] and r['Counter_Name']==c:
                s+=float(r['Counter_Value']); n+=1
    return s,n
f,n=tot('fetch','FETCH_SIZE'); w,_=tot('write','WRITE_SIZE')
print('launches',n,'read GB',f*1024*2/1e9,'write GB',w*1024/1e9)
PY
