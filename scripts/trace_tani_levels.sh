cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/tt
rocprofv3 --kernel-trace --output-format csv -d /tmp/tt -o run -- python3 $GRAFT_REPO_ROOT/bench.py --config 5 --steps 3 --warmup 1 --no-cpu-baseline > /tmp/tt.log 2>&1
python3 - <<PY
import csv
rows=[r for r in csv.DictReader(open('/tmp/tt/run_kernel_trace.csv')) if 'tani' in r['Kernel_Name'] or 'sum_partials' in r['Kernel_Name']]
rows.sort(key=lambda r:int(r['Start_Timestamp']))
t=[r for r in rows if 'tani' in r['Kernel_Name']]
print('tani  ', [(r['Grid_Size_X'] if 'Grid_Size_X' in r else r.get('Grid_Size'), round((int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3,1)) for r in t[12:24]])
s=[r for r in rows if 'sum_partials' in r['Kernel_Name']]
print('sum_p ', [round((int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3,1) for r in s[11:22]])
PY
