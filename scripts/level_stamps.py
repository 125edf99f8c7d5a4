"""Where a level launch's time goes, by in-kernel s_memrealtime stamps (diagnostic build -DLW_STAMPS of
level_reduce_mfma.hip: make -C sober_amd/csrc BUILD=build_lws EXTRA='-DSOBER_DIAG_BUILD -DLW_STAMPS' OUT=build_lws/libsober_hip_lws.so):
   SOBER_HIP_LIB=.../libsober_hip_lws.so python scripts/level_stamps.py [d=10] [rows=700] [S=200] [pool=100000] -- n ...
For every n (live positions of the launch): the stamps of all waves relative to the earliest wave's entry, in microseconds -- 0 entry | 1 after the queued-level block | 2 after the table barrier | 3 row fragments arrived |
4 first two candidates arrived | 5 element loop done | 6 after the LDS barrier | 7 stores issued."""
import os as _os; _os.environ.setdefault("SOBER_ALLOW_DIAG_LIB", "1")   # (a stamped library is a diagnostic build)
import os, sys, json
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sober_amd import _native as nat

args = sys.argv[1:]
k = args.index("--") if "--" in args else len(args)
kv = dict(a.split("=") for a in args[:k])
d, n_rows, S, pool = int(kv.get("d", 10)), int(kv.get("rows", 700)), int(kv.get("S", 200)), int(kv.get("pool", 100000))
sizes = [int(x) for x in args[k + 1:]] or [100000, 50000, 25000, 12500, 6250, 3125, 1600, 800, 400]
dev = torch.device("cuda")
g = torch.Generator().manual_seed(0)
X = torch.rand(pool, d, generator=g, dtype=torch.float64).to(dev)
R = torch.rand(n_rows, d, generator=g, dtype=torch.float64).to(dev)
ls = torch.full((1,), 0.7 * d ** 0.5, dtype=torch.float64, device=dev)
center = X.mean(0)
da = nat.load().sober_aug_dim(d)
cand = torch.empty(pool, da, dtype=torch.float64, device=dev)
rows = torch.empty(n_rows, da, dtype=torch.float64, device=dev)
nat.augment_points(R, ls, center, 0, rows)
nat.augment_points(X, ls, center, 1, cand)
mu = torch.rand(pool, generator=g, dtype=torch.float64).to(dev)
partG = torch.zeros(8 * n_rows * S, dtype=torch.float64, device=dev)
partTot = torch.empty(8 * S, dtype=torch.float64, device=dev)
other = torch.rand(2048, 2048, device=dev)
clk = None
names = ["entry", "queued", "table+bar", "rows in", "cands in", "loop done", "lds bar", "stores out"]
for n in sizes:
    idx = torch.randperm(pool, generator=g)[:n].to(torch.int32).to(dev)
    P = nat.level_parts_mfma(n_rows, 0, n, S)
    E = n // S
    def run():
        nat.level_reduce_mfma(0, rows, cand, da, idx, 0, 0, n, S, mu, None, 1.3, P, partG, S, 0, partTot, E * S)
    for _ in range(3):
        run()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    torch.cuda.synchronize()
    ev[0].record()
    for _ in range(20):
        run()
    ev[1].record()
    torch.cuda.synchronize()
    us_back_to_back = ev[0].elapsed_time(ev[1]) * 1e3 / 20
    for mode in ("warm", "after another kernel"):
        partG[6 * n_rows * S:].zero_()
        if mode != "warm":
            (other @ other).sum().item()
        torch.cuda.synchronize()
        run()
        torch.cuda.synchronize()
        st = partG[6 * n_rows * S:].view(torch.int64).cpu().numpy()
        st = st[: (st.size // 8) * 8].reshape(-1, 8)
        import numpy as np
        wg = np.arange(len(st)) // 4
        keepm = st[:, 0] != 0
        if os.environ.get("LW_DUMP") and mode == "warm":
            np.save(f"gpurun_out/level_stamps_{n}.npy", st)
        st = st[keepm]
        t0 = st[:, 0].min()
        if mode == "warm":                                   # loop duration and finish by XCD (workgroup id & 7)
            x = (wg[keepm] & 7)
            dur = (st[:, 5] - st[:, 4]) / 100.0
            fin = (st[:, 5] - t0) / 100.0
            ok = st[:, 5] != 0
            print(json.dumps({"n": n, "loop_us_by_xcd[med,max]": [[round(float(np.median(dur[ok & (x == k)])), 1), round(float(dur[ok & (x == k)].max()), 1)] for k in range(8)],
                              "finish_by_xcd[med,max]": [[round(float(np.median(fin[ok & (x == k)])), 1), round(float(fin[ok & (x == k)].max()), 1)] for k in range(8)],
                              "loop_us_percentiles[5,25,50,75,95,100]": [round(float(v), 1) for v in np.percentile(dur[ok], [5, 25, 50, 75, 95, 100])]}), flush=True)
        clk = 100.0                                         # s_memrealtime: 100 MHz, one origin for the whole chip
        line = {"n": n, "mode": mode, "waves": int(len(st)), "P": P, "back_to_back_us": round(us_back_to_back, 2)}
        for j, nm in enumerate(names):
            col = st[:, j]
            col = col[col != 0]
            if col.size == 0:
                continue
            rel = (col - t0) / clk
            line[nm] = [round(float(rel.min()), 2), round(float(sorted(rel)[len(rel) // 2]), 2), round(float(rel.max()), 2)]
        print(json.dumps(line), flush=True)
