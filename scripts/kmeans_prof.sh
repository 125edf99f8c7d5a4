# per-kernel durations of scripts/kmeans_time.py (rocprofv3 kernel stats); KM_SIZES="100000:10" restricts the sizes
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/kmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kmp -o run -- python3 $R/scripts/kmeans_time.py > /tmp/kmp.log 2>&1 || tail -5 /tmp/kmp.log
cat /tmp/kmp.log | grep "per KMeans"
python3 - <<PY
import csv
tot=0
for r in csv.DictReader(open('/tmp/kmp/run_kernel_stats.csv')):
    tot+=float(r['TotalDurationNs'])
    if float(r['Percentage']) > ${KM_MINPCT:-0.5}: print('  ', r['Name'][:90], r['Calls'], 'avg us %.1f' % (float(r['AverageNs'])/1e3), 'total ms %.2f' % (float(r['TotalDurationNs'])/1e6), r['Percentage'])
print('total kernel ms', tot/1e6)
PY
