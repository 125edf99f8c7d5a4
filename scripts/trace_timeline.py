"""Timeline of one recombination step from a rocprofv3 kernel trace: start, duration and the idle gap in front of
every kernel (usage: trace_timeline.py <kernel_trace.csv> [step index from the end, default 1])."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
marks = [i for i, r in enumerate(rows) if 'k_abs_sym' in r['Kernel_Name']]
k = int(sys.argv[2]) if len(sys.argv) > 2 else 1
a, b = marks[-k - 1], marks[-k]
t0 = int(rows[a]['Start_Timestamp'])
prev_end, busy = None, 0
for r in rows[a - 12:b - 12]:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    gap = (s - prev_end) / 1e3 if prev_end else 0
    busy += e - s
    name = r['Kernel_Name'].replace('sober::', '').replace('void ', '')[:48]
    print(f"{(s - t0) / 1e3:9.1f} us  dur {(e - s) / 1e3:7.1f}  gap {gap:7.1f}  {name}")
    prev_end = e
print("step span %.1f us, GPU busy %.1f us" % ((prev_end - int(rows[a - 12]['Start_Timestamp'])) / 1e3, busy / 1e3))
