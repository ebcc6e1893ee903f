#!/bin/bash
# Collects the rocprofv3 evidence for one configuration: kernel-trace stats of the bench command,
# then PMC passes (counters only; never combined with tracing domains).
# usage (on the GPU box, from the repo root):
#   bash profiles/run_profiles.sh r02                        # headline config (bench.py defaults)
#   bash profiles/run_profiles.sh r02_c1 "--envs-per-gpu 256 --frame 128 --spp 4" 200
#   (tag, extra bench.py arguments, steps of the traced run)
# then, back in the container:  python profiles/summarize.py <tag>
set -u
#   bash profiles/run_profiles.sh r03_general_mixed "--scene mixed" 0 tools/bench_general.py     # another program
TAG=${1:-r03}
ARGS=${2:-}
TRACE_STEPS=${3:-50}
PROG=${4:-bench.py}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
# kernel trace of the bench command at its default length so that the average launch duration is
# the one bench.py's HIP events report; the PMC passes replay every kernel and use a short run
if [ "$PROG" = "bench.py" ]; then
  BENCH="python3 $ROOT/bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-pmc $ARGS"
  TRACE="python3 $ROOT/bench.py --steps $TRACE_STEPS --no-cpu-baseline --no-pmc $ARGS"
else
  BENCH="python3 $ROOT/$PROG $ARGS"
  TRACE="$BENCH"
fi
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- $TRACE > $OUT/trace.log 2>&1
echo "trace rc=$?"
grep "^{" $OUT/trace.log > $OUT/bench.json
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_INST_ANY \
    --output-format csv -d $OUT/pmc_sq -- $BENCH > $OUT/pmc_sq.log 2>&1
echo "pmc_sq rc=$?"
rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_TRANS GRBM_GUI_ACTIVE GRBM_COUNT \
    --output-format csv -d $OUT/pmc_sq2 -- $BENCH > $OUT/pmc_sq2.log 2>&1
echo "pmc_sq2 rc=$?"
# dynamic VALU mix by instruction class (the classes issue at different rates on gfx950: tools/ubench/pairbench)
rocprofv3 --pmc SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU \
    --output-format csv -d $OUT/pmc_mix -- $BENCH > $OUT/pmc_mix.log 2>&1
echo "pmc_mix rc=$?"
rocprofv3 --pmc SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_ACTIVE_INST_VALU2 SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE \
    --output-format csv -d $OUT/pmc_mix2 -- $BENCH > $OUT/pmc_mix2.log 2>&1
echo "pmc_mix2 rc=$?"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- $BENCH > $OUT/pmc_fetch.log 2>&1
echo "pmc_fetch rc=$?"
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- $BENCH > $OUT/pmc_write.log 2>&1
echo "pmc_write rc=$?"
find $OUT -name "*.csv" | head -30
