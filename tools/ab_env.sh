# same-box A/B of an environment setting: bash tools/ab_env.sh "MMHIP_NT_RULE=8" [bench args]
SETTING=$1; shift
for i in 1 2 3; do
  for which in base with; do
    if [ $which = with ]; then export $SETTING; else unset ${SETTING%%=*}; fi
    python bench.py --no-cpu-baseline "$@" 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$which', d['ms_per_step'], d['fwd_bwd_ms'], d['roofline']['achieved'])"
  done
done
