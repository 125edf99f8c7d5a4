import ctypes as C, os, sys, json
import numpy as np, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from sober_amd import _native as nat
dev = torch.device("cuda")
lib = nat.load(); st = torch.cuda.current_stream().cuda_stream
g = torch.Generator().manual_seed(0)
n_rows, S, dt = 700, 200, 32
for n in (250000, 125000, 62500, 31250, 7800):
    cand = torch.randint(-2**62, 2**62, (n, dt), generator=g, dtype=torch.int64).to(dev)
    rows = torch.randint(-2**62, 2**62, (n_rows, dt), generator=g, dtype=torch.int64).to(dev)
    cn = torch.full((n,), 1000.0, dtype=torch.float64, device=dev); rn = torch.full((n_rows,), 1000.0, dtype=torch.float64, device=dev)
    idx = torch.arange(n, dtype=torch.int32, device=dev)
    mu = torch.rand(n, generator=g, dtype=torch.float64).to(dev)
    partG = torch.zeros(64 * n_rows * S, dtype=torch.float64, device=dev); partTot = torch.zeros(64 * S, dtype=torch.float64, device=dev)
    res = {}
    for P in (3, 4, 5, 6, 8, 12, 13, 19, 26):
        if P > n // S: continue
        def run():
            rc = lib.sober_level_reduce_tani(rows.data_ptr(), rn.data_ptr(), n_rows, cand.data_ptr(), cn.data_ptr(), dt, idx.data_ptr(),
                                             0, n, S, mu.data_ptr(), None, 1.3, P, partG.data_ptr(), S, 0, partTot.data_ptr(), (n // S) * S, st)
            assert rc == 0, rc
        for _ in range(3): run()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): run()
        e1.record(); torch.cuda.synchronize()
        res[P] = round(e0.elapsed_time(e1) * 100, 1)
    print(n, res)
