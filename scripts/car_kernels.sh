# per-kernel durations of one CAR call (scripts/car_time.py) for the current library and, if given, a second one
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for lib in "" "$1"; do
  if [ -n "$lib" ]; then export SOBER_HIP_LIB=$R/$lib; fi
  rm -rf /tmp/cark
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/cark -o run -- python3 $R/scripts/car_time.py > /tmp/cark.log 2>&1 || tail -5 /tmp/cark.log
  echo "lib: ${lib:-default}"
  python3 - <<PY
import csv
for r in csv.DictReader(open('/tmp/cark/run_kernel_stats.csv')):
    if 'car' in r['Name']: print('  ', r['Name'][:48], r['Calls'], 'avg us', float(r['AverageNs'])/1e3, 'min', float(r['MinNs'])/1e3)
PY
  [ -z "$1" ] && break
done
