#!/bin/bash
# round 4, call 12: where the LDS bank-conflict cycles of the headline instance come from -- counter builds (wrong
# frames, counters only): 1 = round-2 results not scattered, 2 = no same-address atomics in round 1, 4 = colour sums out of LDS
set -u
OUT=$PWD/gpurun_out/r04_l; mkdir -p $OUT; rm -rf $OUT/*
ROOT=$PWD
cd /tmp && export TMPDIR=/tmp
for lib in reinfocus_amd/libreinfocus_hip.so tools/lib_lds1.so tools/lib_lds2.so tools/lib_lds4.so; do
  tag=$(basename $lib .so)
  REINFOCUS_HIP_LIB=$ROOT/$lib rocprofv3 --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_WAVES SQ_INSTS_VALU --output-format csv -d $OUT/$tag -- python3 $ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-pmc > $OUT/$tag.log 2>&1
  echo "$tag rc=$?"
  python3 - $OUT/$tag <<'PY'
import csv,glob,sys,collections
t=collections.defaultdict(float)
for f in glob.glob(sys.argv[1]+'/*/*_counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        if 'coop2' in r['Kernel_Name']: t[r['Counter_Name']]+=float(r['Counter_Value'])
w=t['SQ_WAVES']*16
print({k: round(v/w,2) for k,v in t.items()}, 'conflict share', round(t['SQ_LDS_BANK_CONFLICT']/max(t['SQ_LDS_IDX_ACTIVE'],1),4))
PY
done
