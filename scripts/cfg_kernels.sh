# kernel-time table of one bench configuration (per step), rocprofv3 --stats:  bash scripts/cfg_kernels.sh 2
R=$GRAFT_REPO_ROOT
C=${1:-2}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/ck
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ck -o run -- python3 $R/bench.py --config $C --steps 10 --warmup 3 --no-cpu-baseline --no-sweep --no-others > /tmp/ck.log 2>&1 || tail -5 /tmp/ck.log
python3 - <<PY
import csv, json
rows = list(csv.DictReader(open('/tmp/ck/run_kernel_stats.csv')))
# steps of the whole run (initialisation, cold, warm-up, timed, parity): one k_car_pivot_stream / k_mc_pivot per level, counted through the final-level kernel
fin = [r for r in rows if 'k_abs_sym' in r['Name']]
steps = float(fin[0]['Calls']) if fin else 14.0
tot = sum(float(r['TotalDurationNs']) for r in rows)
print("kernel time per step (all %d steps of the run averaged): %.3f ms" % (steps, tot / steps / 1e6))
for r in rows[:70]:
    print("  %-64s calls/step %5.1f  avg us %8.2f  ms/step %.3f" % (r['Name'][:64], float(r['Calls']) / steps, float(r['AverageNs']) / 1e3, float(r['TotalDurationNs']) / steps / 1e6))
PY
grep '^{' /tmp/ck.log | tail -1 | python3 -c "import sys, json; d = json.loads(sys.stdin.read()); print('ms_per_step', d['ms_per_step'], d.get('phases_ms_per_step'))"
