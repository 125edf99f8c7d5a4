# per-kernel durations of scripts/kmeans_time.py (rocprofv3 kernel stats)
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/kmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kmp -o run -- python3 $R/scripts/kmeans_time.py > /tmp/kmp.log 2>&1 || tail -5 /tmp/kmp.log
cat /tmp/kmp.log | grep "per KMeans"
python3 - <<PY
import csv
for r in csv.DictReader(open('/tmp/kmp/run_kernel_stats.csv')):
    if float(r['Percentage']) > 0.5: print('  ', r['Name'][:70], r['Calls'], 'avg us %.1f' % (float(r['AverageNs'])/1e3), 'min %.1f' % (float(r['MinNs'])/1e3), 'max %.1f' % (float(r['MaxNs'])/1e3), r['Percentage'])
PY
