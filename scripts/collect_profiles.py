#!/usr/bin/env python3
"""Copy the judged summaries of one scripts/refresh_profiles.sh run (gpurun_out/<tag>) into profiles/ (tracked), named
per round:  collect_profiles.py <tag> <rNN>
  <rNN>_bench_cfgC.json          the bench line of configuration C (driver-shaped command)
  <rNN>_cfgC_kernel_stats.csv    rocprofv3 --kernel-trace --stats of `bench.py --config C --steps 10 --warmup 3`
  <rNN>_pmc_cfgC.csv             FETCH_SIZE / WRITE_SIZE per dispatch of the level kernels (separate passes)
  <rNN>_pmc_traffic.json         per configuration: HBM bytes per launch of the dominant kernel =
                                 (2 * mean FETCH_SIZE + mean WRITE_SIZE) * 1024 -- FETCH_SIZE counts 64 B per 128-B
                                 request on gfx950 (MI355X_MICROARCH.md, HBM section); bench.py reads this file
  <rNN>_pmc_sq*_level_reduce.csv SQ counters per dispatch of the level kernel, last step (configuration 2)
  <rNN>_dp_rate.txt              scripts/dp_rate_probe: what the FP64 units sustain chip-wide
  <rNN>_roofline_check.txt       bench.py's kernel_ms_per_step against calls x AverageNs / steps of the stats run
"""
import csv
import json
import shutil
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
tag = sys.argv[1] if len(sys.argv) > 1 else "r04f"
name = sys.argv[2] if len(sys.argv) > 2 else "r04"
src = ROOT / "gpurun_out" / tag
dst = ROOT / "profiles"
DOMINANT = ("k_level_reduce_wave", "k_level_reduce_tani", "k_level_reduce<", "k_level_gather")

traffic, check = {}, []
for c in range(1, 6):
    b = src / f"bench_cfg{c}.json"
    line = None
    if b.exists():
        lines = [ln for ln in b.read_text().splitlines() if ln.startswith("{")]
        if lines:
            line = json.loads(lines[-1])
            (dst / f"{name}_bench_cfg{c}.json").write_text(lines[-1] + "\n")
    k = src / f"prof_cfg{c}" / "run_kernel_stats.csv"
    if k.exists():
        shutil.copy(k, dst / f"{name}_cfg{c}_kernel_stats.csv")
        tot_ns = calls = 0
        for r in csv.DictReader(k.open()):
            if any(d in r["Name"] for d in DOMINANT):
                tot_ns += float(r["TotalDurationNs"]) if "TotalDurationNs" in r else float(r["AverageNs"]) * int(r["Calls"])
                calls += int(r["Calls"])
        # the stats run: 1 initialisation + 3 warm-up + 10 timed steps = 14 steps of the same work (+ 6 steps with a fresh
        # pool for fingerprint pools: ms_per_step_fresh_pool) -- the step count follows from the launches per step
        if line is not None and calls:
            n_steps = max(1, round(calls / line["roofline"]["launches_per_step"]))
            per_step = tot_ns / n_steps / 1e6
            mine = line["roofline"]["kernel_ms_per_step"]
            check.append(f"cfg {c}: rocprofv3 {calls} calls, {per_step:.4f} ms/step ({n_steps} steps); bench.py (its own run) "
                         f"{mine:.4f} ms/step over {line['roofline']['launches_per_step']:.1f} launches/step; "
                         f"ratio {mine / per_step:.3f}")
    rows, acc = [], {}
    for pas, counter in (("fetch", "FETCH_SIZE"), ("write", "WRITE_SIZE")):
        cc = src / f"pmc_{pas}_cfg{c}" / "run_counter_collection.csv"
        if not cc.exists():
            continue
        for r in csv.DictReader(cc.open()):
            kn = r["Kernel_Name"]
            if "k_level_reduce" not in kn and "k_sum_partials" not in kn and "k_level_gather" not in kn:
                continue
            val = float(r["Counter_Value"])
            dur = int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) if "End_Timestamp" in r else ""
            rows.append((pas, r["Dispatch_Id"], kn.split("(")[0], r["Grid_Size"], counter, f"{val:.6f}", dur))
            if any(d in kn for d in DOMINANT):
                a = acc.setdefault(pas, [0.0, 0])
                a[0] += val
                a[1] += 1
    if rows:
        with (dst / f"{name}_pmc_cfg{c}.csv").open("w", newline="") as f:
            w = csv.writer(f)
            w.writerow(["pass", "Dispatch_Id", "Kernel_Name", "Grid_Size", "Counter_Name", "Counter_Value_KB", "duration_ns"])
            w.writerows(rows)
    if "fetch" in acc and "write" in acc:
        fe, wr = acc["fetch"], acc["write"]
        traffic[str(c)] = {"bytes_per_launch": (2.0 * fe[0] / fe[1] + wr[0] / wr[1]) * 1024.0,
                           "fetch_kb_mean": fe[0] / fe[1], "write_kb_mean": wr[0] / wr[1],
                           "launches_fetch_pass": fe[1], "launches_write_pass": wr[1],
                           "formula": "(2 * FETCH_SIZE + WRITE_SIZE) * 1024 per launch, separate --pmc passes",
                           "source": f"profiles/{name}_pmc_cfg{c}.csv"}
if traffic:
    (dst / f"{name}_pmc_traffic.json").write_text(json.dumps(traffic, indent=1) + "\n")
    print({k: round(v["bytes_per_launch"] / 1e6, 2) for k, v in traffic.items()}, "MB per launch")
if check:
    (dst / f"{name}_roofline_check.txt").write_text("\n".join(check) + "\n")
    print("\n".join(check))
dp = src / "dp_rate.txt"
if dp.exists():
    shutil.copy(dp, dst / f"{name}_dp_rate.txt")

# SQ passes (configuration 2): one row per level-kernel dispatch of the LAST step, counters side by side
for pas in ("sq", "sq2"):
    cc = src / f"pmc_{pas}" / "run_counter_collection.csv"
    if not cc.exists():
        continue
    per, order, names = {}, [], []
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
    big = max(per[d]["dur"] for d in order)
    last0 = max(i for i, d in enumerate(order) if per[d]["dur"] >= 0.7 * big)
    with (dst / f"{name}_pmc_{pas}_level_reduce.csv").open("w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["dispatch", "grid", "duration_us"] + names)
        for d in order[last0:]:
            w.writerow([d, per[d]["grid"], f"{per[d]['dur'] / 1e3:.1f}"] + [f"{per[d].get(n, float('nan')):.6g}" for n in names])
