# debug + timing of the Gram route of the Caratheodory step on the GPU box
R=$GRAFT_REPO_ROOT
cd $R
timeout 300 python3 scripts/car_gram_debug.py "$@" 2>&1 | tail -60
echo "--- timing (Gram route)"
timeout 120 python3 scripts/car_time.py 2>&1 | tail -2
echo "--- timing (bidiagonalisation, SOBER_CAR_NO_GRAM=1)"
SOBER_CAR_NO_GRAM=1 timeout 120 python3 scripts/car_time.py 2>&1 | tail -2
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/cark
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/cark -o run -- python3 $R/scripts/car_time.py > /tmp/cark.log 2>&1 || tail -5 /tmp/cark.log
python3 - <<PY
import csv
for r in csv.DictReader(open('/tmp/cark/run_kernel_stats.csv')):
    if 'car' in r['Name']: print('  ', r['Name'][:60], r['Calls'], 'avg us', float(r['AverageNs'])/1e3, 'min', float(r['MinNs'])/1e3)
PY
