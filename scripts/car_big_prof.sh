# kernel-time table of the memory-resident Caratheodory route at one size:  bash scripts/car_big_prof.sh 500 251
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/cb
cat > /tmp/cb_run.py <<PY
import sys, numpy as np, torch
sys.path.insert(0, "$R")
from sober_amd import _native as nat
dev = torch.device("cuda:0")
N, m = ${1:-500}, ${2:-251}
rng = np.random.default_rng(N)
X = torch.from_numpy(rng.standard_normal((N, m - 1))).to(dev)
mu = torch.from_numpy(rng.random(N) + 0.05).to(dev)
kr = torch.empty(N, dtype=torch.int32, device=dev); ws = torch.zeros(N, dtype=torch.float64, device=dev)
nk = torch.empty(1, dtype=torch.int32, device=dev); mo = torch.empty(N, dtype=torch.float64, device=dev)
for rep in range(5):
    nat.car_device(X, mu, kr, ws, nk, mo, big=True)
torch.cuda.synchronize()
PY
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/cb -o run -- python3 /tmp/cb_run.py > /tmp/cb.log 2>&1 || tail -5 /tmp/cb.log
python3 - <<PY
import csv
rows = list(csv.DictReader(open('/tmp/cb/run_kernel_stats.csv')))
for r in rows[:12]:
    print("  %-60s calls/step %7.1f  avg us %8.2f  ms/step %.3f" % (r['Name'][:60], float(r['Calls']) / 5, float(r['AverageNs']) / 1e3, float(r['TotalDurationNs']) / 5 / 1e6))
PY
