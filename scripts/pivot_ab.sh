# screened vs exact ratio test of the pivot stream: per-kernel durations of one 200 x 100 step (scripts/car_time.py)
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for sw in "" 1; do
  if [ -n "$sw" ]; then export SOBER_CAR_EXACT_RATIO=1; fi
  rm -rf /tmp/cark
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/cark -o run -- python3 $R/scripts/car_time.py > /tmp/cark.log 2>&1 || tail -5 /tmp/cark.log
  echo "SOBER_CAR_EXACT_RATIO=${sw:-off}: $(tail -1 /tmp/cark.log)"
  python3 - <<PY
import csv
for r in csv.DictReader(open('/tmp/cark/run_kernel_stats.csv')):
    if 'car' in r['Name']: print('  ', r['Name'][:48], r['Calls'], 'avg us', float(r['AverageNs'])/1e3, 'min', float(r['MinNs'])/1e3)
PY
done
