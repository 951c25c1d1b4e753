# same-box A/B of two builds: MMHIP_LIB_PATH=<old .so> vs the in-tree library, alternating, full step + fwd/bwd
OLD=$1; shift
for i in 1 2 3; do
  for which in old new; do
    if [ $which = old ]; then export MMHIP_LIB_PATH=$OLD; else unset MMHIP_LIB_PATH; fi
    python bench.py --no-cpu-baseline "$@" 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$which', d['ms_per_step'], d['fwd_bwd_ms'], d['roofline']['achieved'])"
  done
done
