cd /tmp && export TMPDIR=/tmp
for v in mcNOWAIT mcNOPROG mcBOTH; do
  export SOBER_HIP_LIB=$GRAFT_REPO_ROOT/sober_amd/csrc/build_$v/libsober_hip_$v.so
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/carmc_$v -o run -- python3 $GRAFT_REPO_ROOT/scripts/car_mc_time.py > /dev/null 2>&1
  python3 - <<PY
import csv
for r in csv.DictReader(open('/tmp/carmc_$v/run_kernel_stats.csv')):
    if 'k_mc_bidiag' in r['Name']: print('$v', r['Name'][:34], r['Calls'], 'avg us', round(float(r['AverageNs'])/1e3,1))
PY
done
