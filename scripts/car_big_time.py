import sys, time, numpy as np, torch
sys.path.insert(0, "/root/repo")
from sober_amd import _native as nat
dev = torch.device("cuda:0")
for (N, m) in [(400, 201), (448, 225), (500, 251), (600, 301), (800, 401), (1024, 513), (2048, 1025)]:
    rng = np.random.default_rng(N)
    X = torch.from_numpy(rng.standard_normal((N, m - 1))).to(dev)
    mu = torch.from_numpy(rng.random(N) + 0.05).to(dev)
    kr = torch.empty(N, dtype=torch.int32, device=dev); ws = torch.zeros(N, dtype=torch.float64, device=dev)
    nk = torch.empty(1, dtype=torch.int32, device=dev); mo = torch.empty(N, dtype=torch.float64, device=dev)
    for kw in (dict(big=True), dict(multi_cu=True)):
        if "multi_cu" in kw and N > 448: continue
        for rep in range(3):
            torch.cuda.synchronize(); t = time.perf_counter()
            nat.car_device(X, mu, kr, ws, nk, mo, **kw)
            t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
        print(N, m, kw, "enqueue %.2f ms  total %.2f ms  n_keep %d" % ((t1 - t) * 1e3, (t2 - t) * 1e3, int(nk.item())), flush=True)
    t = time.perf_counter()
    A = torch.cat([torch.ones(N, 1, dtype=torch.float64), X.cpu()], 1)
    torch.linalg.svd(A.T)
    print("   host svd %.1f ms" % ((time.perf_counter() - t) * 1e3), flush=True)
