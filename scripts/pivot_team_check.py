"""k_car_pivot_team against k_car_pivot_stream (SOBER_CAR_PIVOT_STREAM=1, read at every call): the same bits in
keep_rank / w_star / n_keep / mu_out on the reference's level inputs and on random steps of every shape the team
kernel takes (N > 128), incl. weights with zeros, columns without a positive entry (Q6) and repeated calls; then
the time per step of both (events around 50 calls)."""
import glob, os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sober_amd import _native as nat
dev = torch.device("cuda:0")


def run(X, mu, stream, mode=nat.CAR_DEFAULT):
    if stream: os.environ["SOBER_CAR_PIVOT_STREAM"] = "1"
    else: os.environ.pop("SOBER_CAR_PIVOT_STREAM", None)
    N = X.shape[0]
    Xd, mud = torch.from_numpy(np.ascontiguousarray(X)).to(dev), torch.from_numpy(mu).to(dev)
    kr = torch.full((N,), -7, dtype=torch.int32, device=dev); ws = torch.zeros(N, dtype=torch.float64, device=dev)
    nk = torch.full((1,), -9, dtype=torch.int32, device=dev); mo = torch.zeros(N, dtype=torch.float64, device=dev)
    nat.car_device(Xd, mud, kr, ws, nk, mo, mode=mode)
    torch.cuda.synchronize()
    return kr.cpu().numpy(), ws.cpu().numpy(), int(nk.item()), mo.cpu().numpy()


def same(a, b):
    return a[2] == b[2] and np.array_equal(a[0], b[0]) and np.array_equal(a[1].view(np.int64), b[1].view(np.int64)) \
        and np.array_equal(a[3].view(np.int64), b[3].view(np.int64))


bad = 0
cases = []
gold = os.path.join(os.path.dirname(__file__), "..", "tests", "golden")
for f in sorted(glob.glob(os.path.join(gold, "recomb_*.npz"))):
    z = np.load(f)
    for k in z.files:
        if k.endswith("_X_tmp") and z[k].shape[0] > 128 and z[k].shape[0] <= 208:
            cases.append((os.path.basename(f) + ":" + k, z[k], z[k.replace("_X_tmp", "_tot_weights")]))
rng = np.random.default_rng(7)
for (N, m) in [(200, 100), (200, 101), (199, 100), (208, 112), (208, 96), (130, 100), (129, 112), (150, 100), (192, 100),
               (193, 100), (160, 112), (201, 100), (140, 30), (131, 129 - 17)]:
    if m > 112 or N - m > 112 or N > 208: continue
    for rep in range(3):
        X = rng.standard_normal((N, m - 1)) * np.exp(rng.uniform(-6, 0, size=(1, m - 1)))
        mu = rng.uniform(0.1, 1.0, N); mu /= mu.sum()
        if rep == 1: mu[rng.integers(0, N, 5)] = 0.0
        cases.append(("rand N=%d m=%d r%d" % (N, m, rep), X, mu))
for name, X, mu in cases:
    a, b = run(X, mu, False), run(X, mu, True)
    a2 = run(X, mu, False)
    ok = same(a, b) and same(a, a2)
    if not ok:
        bad += 1
        print("MISMATCH", name, X.shape, "n_keep team/stream", a[2], b[2], "kr equal", np.array_equal(a[0], b[0]),
              "max |dw|", float(np.abs(a[3] - b[3]).max()))
    else:
        print("ok", name, X.shape, "n_keep", a[2])
# unfused / safe route (no err words) on one case
name, X, mu = cases[0]
a, b = run(X, mu, False, nat.CAR_SAFE), run(X, mu, True, nat.CAR_SAFE)
print("safe route", "ok" if same(a, b) else "MISMATCH")
bad += 0 if same(a, b) else 1
print("cases", len(cases), "mismatches", bad)
# timing
z = np.load(os.path.join(gold, "recomb_matern_medium.npz"))
X, mu = np.ascontiguousarray(z["L0_X_tmp"]), z["L0_tot_weights"]
N = X.shape[0]
Xd, mud = torch.from_numpy(X).to(dev), torch.from_numpy(mu).to(dev)
kr = torch.empty(N, dtype=torch.int32, device=dev); ws = torch.empty(N, dtype=torch.float64, device=dev)
nk = torch.empty(1, dtype=torch.int32, device=dev); mo = torch.empty(N, dtype=torch.float64, device=dev)
for rep in range(2):
    for stream in (False, True):
        if stream: os.environ["SOBER_CAR_PIVOT_STREAM"] = "1"
        else: os.environ.pop("SOBER_CAR_PIVOT_STREAM", None)
        for it in range(5): nat.car_device(Xd, mud, kr, ws, nk, mo)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for it in range(50): nat.car_device(Xd, mud, kr, ws, nk, mo)
        e1.record(); torch.cuda.synchronize()
        print("stream" if stream else "team  ", "avg ms per sober_car_device", e0.elapsed_time(e1) / 50, "n_keep", int(nk.item()))
sys.exit(1 if bad else 0)
