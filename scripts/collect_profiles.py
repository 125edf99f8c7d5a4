#!/usr/bin/env python3
"""Copy the judged summaries of one scripts/refresh_profiles.sh run (gpurun_out/<tag>) into profiles/ (tracked):
bench lines, rocprofv3 kernel stats per configuration, and one CSV with the FETCH_SIZE / WRITE_SIZE passes of the level
kernels (per dispatch, with the dispatch's duration from the same pass's kernel trace).  Prints the per-launch HBM traffic
of the level kernel (2 * FETCH_SIZE + WRITE_SIZE: the gfx950 correction of MI355X_MICROARCH.md's HBM section)."""
import csv
import shutil
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
tag = sys.argv[1] if len(sys.argv) > 1 else "r02f"
name = sys.argv[2] if len(sys.argv) > 2 else "r02"
src = ROOT / "gpurun_out" / tag
dst = ROOT / "profiles"

for c in range(1, 6):
    b = src / f"bench_cfg{c}.json"
    if b.exists():
        lines = [ln for ln in b.read_text().splitlines() if ln.startswith("{")]
        (dst / f"{name}_bench_cfg{c}.json").write_text(lines[-1] + "\n")
    k = src / f"prof_cfg{c}" / "run_kernel_stats.csv"
    if k.exists():
        shutil.copy(k, dst / f"{name}_cfg{c}_kernel_stats.csv")

rows, per_kernel = [], {}
for pas, counter in (("fetch", "FETCH_SIZE"), ("write", "WRITE_SIZE")):
    cc = src / f"pmc_{pas}" / "run_counter_collection.csv"
    kt = src / f"pmc_{pas}" / "run_kernel_trace.csv"
    if not cc.exists():
        continue
    dur = {}
    for r in csv.DictReader(kt.open()):
        dur[r["Dispatch_Id"]] = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    for r in csv.DictReader(cc.open()):
        kn = r["Kernel_Name"]
        if "k_level_reduce" not in kn and "k_sum_partials" not in kn:
            continue
        kn = kn.split("(")[0]
        val = float(r["Counter_Value"])
        rows.append((pas, r["Dispatch_Id"], kn, r["Grid_Size"], counter, f"{val:.6f}", dur.get(r["Dispatch_Id"], "")))
        if "k_level_reduce" in kn:
            acc = per_kernel.setdefault(pas, [0.0, 0])
            acc[0] += val
            acc[1] += 1
# SQ passes: one row per level-kernel dispatch of the LAST step, counters side by side
for pas in ("sq", "sq2"):
    cc = src / f"pmc_{pas}" / "run_counter_collection.csv"
    if not cc.exists():
        continue
    per = {}
    order = []
    names = []
    for r in csv.DictReader(cc.open()):
        if "k_level_reduce" not in r["Kernel_Name"]:
            continue
        d = r["Dispatch_Id"]
        if d not in per:
            per[d] = {"grid": r["Grid_Size"], "dur": int(r["End_Timestamp"]) - int(r["Start_Timestamp"])}
            order.append(d)
        per[d][r["Counter_Name"]] = float(r["Counter_Value"])
        if r["Counter_Name"] not in names:
            names.append(r["Counter_Name"])
    # the last step = the dispatches from the last level-0 launch on (level 0 = the longest launches of the run)
    big = max(per[d]["dur"] for d in order)
    last0 = max(i for i, d in enumerate(order) if per[d]["dur"] >= 0.7 * big)
    with (dst / f"{name}_pmc_{pas}_level_reduce.csv").open("w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["dispatch", "grid", "duration_us"] + names)
        for d in order[last0:]:
            w.writerow([d, per[d]["grid"], f"{per[d]['dur'] / 1e3:.1f}"] + [f"{per[d].get(n, float('nan')):.6g}" for n in names])
if rows:
    with (dst / f"{name}_pmc_level_reduce.csv").open("w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["pass", "Dispatch_Id", "Kernel_Name", "Grid_Size", "Counter_Name", "Counter_Value_KB", "duration_ns"])
        w.writerows(rows)
    fe = per_kernel.get("fetch", [0, 1])
    wr = per_kernel.get("write", [0, 1])
    per_launch = (2.0 * fe[0] / max(fe[1], 1) + wr[0] / max(wr[1], 1)) * 1024.0
    print(f"level kernel launches: fetch pass {fe[1]}, write pass {wr[1]}; "
          f"HBM traffic per launch = {per_launch / 1e6:.2f} MB (2*FETCH + WRITE)")
