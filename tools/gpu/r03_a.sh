#!/bin/bash
# round 3, GPU call A: diagnostics (counter list, wave -> SIMD mapping), parity + A/B of RF_ROT / RF_NEVERHIT
set -u
OUT=gpurun_out/r03_a; mkdir -p $OUT
export TMPDIR=/tmp
(cd /tmp && rocprofv3 -L > $GRAFT_REPO_ROOT/$OUT/counters.txt 2>&1; echo "rocprofv3 -L rc=$?")
./tools/ubench/hwid > $OUT/hwid.txt 2>&1; echo "hwid rc=$?"; tail -8 $OUT/hwid.txt
for v in rot1 nh1 rot1nh1; do
  REINFOCUS_HIP_LIB=$PWD/tools/lib_$v.so timeout -k 10 300 python -m pytest tests -m gpu -x -q > $OUT/pytest_$v.log 2>&1
  echo "pytest $v rc=$? $(tail -1 $OUT/pytest_$v.log)"
done
REPS=2 bash tools/ab.sh tools/lib_base.so tools/lib_rot1.so tools/lib_nh1.so tools/lib_rot1nh1.so 2>&1 | tee $OUT/ab.log
# derived / extra counters on the base build (counters only)
cd /tmp
for grp in "VALUBusy SALUBusy" "SQ_INST_CYCLES_VMEM SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS"; do
  tag=$(echo $grp | tr ' ' '_' | cut -c1-40)
  rocprofv3 --pmc $grp --output-format csv -d $GRAFT_REPO_ROOT/$OUT/pmc_$tag -- python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-pmc > $GRAFT_REPO_ROOT/$OUT/pmc_$tag.log 2>&1
  echo "pmc [$grp] rc=$?"
done
