# same-box A/B of two checkouts of this repository (each with its own built library): bench.py of the working tree
# against bench.py of <dir>, alternately, `reps` times, for the given configurations
#   bash scripts/ab_tree.sh ab_old "1 2" 2
D=$1; CFGS=${2:-"2"}; REPS=${3:-2}
for c in $CFGS; do for r in $(seq $REPS); do for t in . $D; do
  (cd $t && python3 bench.py --config $c --steps 20 --warmup 5 --no-cpu-baseline --no-sweep 2>/dev/null | python3 -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('cfg $c tree $t ms/step %.3f' % d['ms_per_step'], {k: round(v,3) for k,v in d['phases_ms_per_step'].items()}, 'kernel %.4f' % d['roofline']['kernel_ms_per_step'])")
done; done; done
