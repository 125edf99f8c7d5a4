# same-box A/B of an environment switch on bench.py:  bash scripts/ab_env.sh VAR [config] [rounds]
# alternates `VAR=1 python bench.py` and `python bench.py` (no CPU baseline, no sweep) and prints ms_per_step + phases
VAR=$1; CFG=${2:-2}; ROUNDS=${3:-3}
cd $GRAFT_REPO_ROOT
for r in $(seq 1 $ROUNDS); do
  for mode in on off; do
    if [ $mode = on ]; then export $VAR=1; else unset $VAR; fi
    python bench.py --config $CFG --steps 20 --warmup 5 --no-cpu-baseline --no-sweep 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$VAR=$mode', 'ms/step %.3f median %.3f cold %.3f' % (d['ms_per_step'], d['ms_per_step_median'], d['ms_per_step_cold']), {k: round(v,3) for k,v in d['phases_ms_per_step'].items()}, d['parity'])"
  done
done
