#!/bin/bash
# rehearsal of bench.py's N > 1 code path on a one-GPU box: two ranks share the card, gloo moves the device tensors
# (RCCL needs one GPU per rank).  Numbers mean nothing; the point is that the path runs and rank 0 prints its line.
# Both ranks run under a watchdog that dumps every thread's Python stack and exits after WATCHDOG seconds (default 150): a rehearsal that
# deadlocks says where.  Output: gpurun_out/two_ranks/rank{0,1}.{out,err}
export MASTER_ADDR=127.0.0.1 MASTER_PORT=${PORT:-29555} WORLD_SIZE=2 LOCAL_RANK=0 MMHIP_DIST_BACKEND=gloo
O=${OUT:-gpurun_out/two_ranks}; mkdir -p $O
RUN="import faulthandler, sys, runpy; faulthandler.dump_traceback_later(${WATCHDOG:-150}, exit=True); sys.argv = ['bench.py'] + sys.argv[1:]; runpy.run_path('bench.py', run_name='__main__')"
RANK=1 python -c "$RUN" --gpus 2 --steps 3 --warmup 1 --batch 16 --no-cpu-baseline "$@" > $O/rank1.out 2> $O/rank1.err &
P1=$!
RANK=0 python -c "$RUN" --gpus 2 --steps 3 --warmup 1 --batch 16 --no-cpu-baseline "$@" > $O/rank0.out 2> $O/rank0.err
R=$?
wait $P1
echo "rank0 rc=$R rank1 rc=$?"; tail -c 1500 $O/rank0.out; tail -5 $O/rank0.err | cut -c1-300; tail -5 $O/rank1.err | cut -c1-300
