#!/bin/bash
# same-box A/B of the training step under different kernel-selection environments: tools/ab_step.sh "<bench args>" "ENV=.. ENV=.." "ENV=.." ...
args="$1"; shift
for rep in 1 2; do
  for v in "$@"; do
    line=$(env $v python bench.py $args --no-cpu-baseline 2>/dev/null | tail -1)
    echo "$rep | $v | $(echo "$line" | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], d["fwd_bwd_ms"], d["roofline"]["frac"], d["roofline"]["frac_serial"])')"
  done
done
