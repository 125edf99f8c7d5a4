"""Debugging aid for the screened KMeans E step: labels after one iteration against the exact kernel, over a list of shapes."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sober_amd import _native as nat
lib = nat.load(); dev = torch.device("cuda")
for shape in os.environ.get("KM_SHAPES", "30011,3,77;30011,3,80;30011,3,512;30011,10,77;20000,9,512;20000,8,64;16384,1,33;20000,5,100;20000,12,100;20000,31,64").split(";"):
    N, d, K = [int(v) for v in shape.split(",")]
    rng = np.random.default_rng(7 * N + d)
    X = torch.from_numpy(rng.random((N, d))).to(dev)
    def run(nbytes, iters):
        c = torch.empty(K, d, dtype=torch.float64, device=dev); cl = torch.empty(N, dtype=torch.int32, device=dev)
        ws = torch.zeros(max(nbytes, 8), dtype=torch.uint8, device=dev)
        nat._check(lib.sober_kmeans_lloyd(X.data_ptr(), N, d, K, iters, c.data_ptr(), cl.data_ptr(), ws.data_ptr() if nbytes else None, nbytes, nat._stream(X)), "km")
        return cl.cpu().numpy(), c.cpu().numpy()
    full = int(lib.sober_kmeans_ws_bytes_screened(N, d, K)) or int(lib.sober_kmeans_ws_bytes(N, d, K))
    a, ca = run(full, 1); b, cb = run(0, 1)
    print(shape, "screened" if int(lib.sober_kmeans_stat_offset(N, d, K)) >= 0 else "fp64", "mismatches", int((a != b).sum()))
