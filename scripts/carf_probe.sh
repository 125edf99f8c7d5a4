cd /tmp && export TMPDIR=/tmp
for g in 0 4 28 32 56; do
  export SOBER_CARF_PER_XCD=$g
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/carf_$g -o run -- python3 $GRAFT_REPO_ROOT/scripts/car_time.py > /dev/null 2>&1
  python3 - <<PY
import csv
for r in csv.DictReader(open('/tmp/carf_$g/run_kernel_stats.csv')):
    if 'fused' in r['Name']: print('per_xcd', $g, 'fused avg us', float(r['AverageNs'])/1e3, 'min', float(r['MinNs'])/1e3)
PY
done
