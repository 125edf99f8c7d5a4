# multi-CU Caratheodory kernels: per-kernel durations (scripts/car_mc_time.py), screened vs exact ratio test
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for sw in "" 1; do
  if [ -n "$sw" ]; then export SOBER_CAR_EXACT_RATIO=1; fi
  rm -rf /tmp/cark
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/cark -o run -- python3 $R/scripts/car_mc_time.py > /tmp/cark.log 2>&1 || tail -5 /tmp/cark.log
  echo "SOBER_CAR_EXACT_RATIO=${sw:-off}"; grep "batch\|ms" /tmp/cark.log | head -8
  python3 - <<PY
import csv
for r in csv.DictReader(open('/tmp/cark/run_kernel_stats.csv')):
    if 'mc' in r['Name'] or 'car' in r['Name']: print('  ', r['Name'][:48], r['Calls'], 'avg us', float(r['AverageNs'])/1e3, 'min', float(r['MinNs'])/1e3)
PY
done
