#!/bin/bash
# The GPU-box jobs of this repository as ONE parametrised script (each call of `gpurun -- 'bash tools/gpu/run.sh <job> ...'` is one
# box; output under gpurun_out/<tag>/).  Rounds 3 and 4 kept a script per call (git log -- tools/gpu); what they did is one of:
#
#   suite [pytest args]                 the GPU test suite (default: tests -m gpu), log in gpurun_out/<tag>/pytest.log
#   smoke                               __graft_entry__.smoke()
#   bench [bench.py args]               one bench.py line -> gpurun_out/<tag>/bench.json
#   ab <lib.so> ... [-- bench.py args]  interleaved A/B of builds of the library on bench.py (tools/ab.sh)
#   ab-general <lib.so|env=VALUE> ...   tools/bench_general.py per library (or per environment switch) at SIZE="256 256 16"
#   profile <tag> "<args>" [steps] [prog]   rocprofv3 kernel stats + PMC passes (profiles/run_profiles.sh; then profiles/summarize.py <tag>)
#   guard [lib.so]                      tests/test_gpu_perf_guard.py (optionally on another build; REINFOCUS_PERF_GUARD_RECORD honoured)
#
# TAG=<name> names the output directory (default: the job).  Variant builds: make -C reinfocus_amd/csrc OUT=../../tools/lib_x.so EXTRA=-D...
set -u -o pipefail
job=${1:?job}; shift
out=gpurun_out/${TAG:-$job}
mkdir -p "$out"
case $job in
  suite)
    timeout -k 10 1100 python -m pytest ${@:-tests} -x -q -m gpu > "$out/pytest.log" 2>&1; rc=$?; tail -n 3 "$out/pytest.log"; exit $rc ;;
  smoke)
    timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tee "$out/smoke.log" ;;
  bench)
    timeout -k 10 600 python bench.py "$@" > "$out/bench.json" 2> "$out/bench.err"; rc=$?; cat "$out/bench.json"; exit $rc ;;
  ab)
    libs=(); while [ $# -gt 0 ] && [ "$1" != "--" ]; do libs+=("$1"); shift; done; [ $# -gt 0 ] && shift
    AB_ARGS="$*" REPS=${REPS:-2} bash tools/ab.sh "${libs[@]}" 2>&1 | tee "$out/ab.txt" ;;
  ab-general)
    for rep in $(seq ${REPS:-2}); do for v in "$@"; do
      echo "== $v"
      if [[ $v == *=* ]]; then env "$v" timeout -k 10 300 python tools/bench_general.py ${SIZE:-256 256 16} || exit 1
      else REINFOCUS_HIP_LIB=$PWD/$v timeout -k 10 300 python tools/bench_general.py ${SIZE:-256 256 16} || exit 1; fi
    done; done 2>&1 | tee "$out/bench_general.txt"
    python tools/tab_general.py "$out/bench_general.txt" ;;
  profile)
    bash profiles/run_profiles.sh "$@" > "$out/run_profiles_$1.log" 2>&1; echo "$1: $(grep -c 'rc=0' "$out/run_profiles_$1.log") passes ok" ;;
  guard)
    [ $# -gt 0 ] && export REINFOCUS_HIP_LIB=$PWD/$1
    timeout -k 10 300 python -m pytest tests/test_gpu_perf_guard.py -x -q -m perf > "$out/guard.log" 2>&1; rc=$?; tail -n 4 "$out/guard.log"; exit $rc ;;
  *) echo "unknown job $job"; exit 2 ;;
esac
