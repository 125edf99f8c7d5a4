# per-kernel times of the acquisition step (bench.py --funnel) under rocprofv3
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/fprof
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/fprof -o run -- python3 $R/bench.py --funnel --steps 10 --warmup 3 --no-cpu-baseline > /tmp/fprof.log 2>&1 || tail -5 /tmp/fprof.log
python3 - <<PY
import csv
rows=list(csv.DictReader(open('/tmp/fprof/run_kernel_stats.csv')))
rows.sort(key=lambda r:-float(r['TotalDurationNs']))
tot=sum(float(r['TotalDurationNs']) for r in rows)
print('total kernel ms', tot/1e6)
for r in rows[:28]:
    print('%-70s calls %5s avg us %8.2f total ms %7.3f' % (r['Name'][:70], r['Calls'], float(r['AverageNs'])/1e3, float(r['TotalDurationNs'])/1e6))
PY
