#!/bin/bash
# usage: ab_env.sh "ENV1=a ENV2=b" "ENV1=c" ...   -- same-box A/B of bench.py under different environments (two rounds)
for round in 1 2; do
  for envs in "$@"; do
    out=$(env $envs python bench.py --no-cpu-baseline ${BENCH_ARGS:-} 2>/dev/null | tail -1)
    python - "$round" "$envs" "$out" <<'PY'
import json, sys
d = json.loads(sys.argv[3])
sp = d.get("spans_ms") or {}
r = d.get("roofline") or {}
print(sys.argv[1], "|", sys.argv[2], "|", d["ms_per_step"], d.get("fwd_bwd_ms"), r.get("frac"), r.get("frac_serial"), "| spans", " ".join("%.2f" % v for v in sp.values()), flush=True)
PY
  done
done
