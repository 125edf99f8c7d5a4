# same-box A/B of the one-CU Caratheodory kernels: per-kernel durations of scripts/car_time.py for the in-tree
# library and every library named on the command line (paths relative to the repository root)
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for rep in 1 2; do
for lib in "" "$@"; do
  if [ -n "$lib" ]; then export SOBER_HIP_LIB=$R/$lib; else unset SOBER_HIP_LIB; fi
  rm -rf /tmp/cark
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/cark -o run -- python3 $R/scripts/car_time.py > /tmp/cark.log 2>&1 || tail -5 /tmp/cark.log
  echo "lib: ${lib:-default}  $(grep 'avg ms' /tmp/cark.log)"
  python3 - <<PY
import csv
for r in csv.DictReader(open('/tmp/cark/run_kernel_stats.csv')):
    if 'car' in r['Name']: print('  ', r['Name'][:48], r['Calls'], 'avg us', float(r['AverageNs'])/1e3, 'min', float(r['MinNs'])/1e3)
PY
done
done
