#!/bin/bash
# usage (on the GPU box): tools/trace_step.sh NAME "ENV=.. ENV=.." [bench args]  -> gpurun_out/NAME_step.txt (+ kernel stats csv)
# rocprofv3 gets the program itself after `--`; the environment is exported in this shell first.
NAME=$1; shift; ENVS=$1; shift
R=$GRAFT_REPO_ROOT; mkdir -p $R/gpurun_out
for kv in $ENVS; do export "$kv"; done
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/tr_$NAME
rocprofv3 --kernel-trace --stats -d /tmp/tr_$NAME -o t --output-format csv -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline "$@" > /tmp/tr_$NAME.log 2>&1 || { tail -5 /tmp/tr_$NAME.log; exit 1; }
python3 $R/tools/step_trace.py /tmp/tr_$NAME 6 > $R/gpurun_out/${NAME}_step.txt
python3 $R/tools/timeline.py /tmp/tr_$NAME 6 > $R/gpurun_out/${NAME}_timeline.txt
cp $(find /tmp/tr_$NAME -name "*kernel_stats.csv" | head -1) $R/gpurun_out/${NAME}_kernel_stats.csv
for kv in $ENVS; do unset "${kv%%=*}"; done
