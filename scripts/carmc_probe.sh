cd /tmp && export TMPDIR=/tmp
for u in 0 1; do
  if [ $u = 1 ]; then export SOBER_CAR_UNFUSED=1; else unset SOBER_CAR_UNFUSED; fi
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/carmc_$u -o run -- python3 $GRAFT_REPO_ROOT/scripts/car_mc_time.py > /dev/null 2>&1
  python3 - <<PY
import csv
for r in csv.DictReader(open('/tmp/carmc_$u/run_kernel_stats.csv')):
    if 'k_mc' in r['Name']: print('unfused' if $u else 'fused', r['Name'][:34], r['Calls'], 'avg us', round(float(r['AverageNs'])/1e3,1))
PY
done
