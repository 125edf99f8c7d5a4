"""How far are the BF16-screened values from the exact ones?  Run with SOBER_HIP_LIB pointing at a diagnostic build whose
margin is 2^-KM_MARGIN_LOG2 (scripts/kmeans_margin.sh): ten Lloyd iterations on several pools with the screened E step and
with the exact (x - c)^2 kernel; a label that differs means a point whose two smallest BF16 values were further apart than
the margin although the exact order is the other one -- the values are off by more than half that margin."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sober_amd import _native as nat
lib = nat.load(); dev = torch.device("cuda")
tot_pts = tot_bad = 0
rows = []
for shape in "1000000,20,500;400000,10,500;400000,3,200;300000,31,256;500000,6,64".split(";"):
    N, d, K = [int(v) for v in shape.split(",")]
    for kind in ("uniform", "normal", "offset"):
        rng = np.random.default_rng(N + 7 * d + len(kind))
        X = rng.random((N, d)) if kind != "normal" else rng.standard_normal((N, d))
        if kind == "offset":
            X = 3.0 * X + 1e3
        Xd = torch.from_numpy(X).to(dev)
        def run(nbytes):
            c = torch.empty(K, d, dtype=torch.float64, device=dev); cl = torch.empty(N, dtype=torch.int32, device=dev)
            ws = torch.zeros(max(nbytes, 8), dtype=torch.uint8, device=dev)
            nat._check(lib.sober_kmeans_lloyd(Xd.data_ptr(), N, d, K, 10, c.data_ptr(), cl.data_ptr(), ws.data_ptr() if nbytes else None, nbytes, nat._stream(Xd)), "km")
            off = int(lib.sober_kmeans_stat_offset(N, d, K))
            listed = int(ws[off:off + 4].view(torch.int32).item()) if nbytes else 0
            return cl, listed
        full = int(lib.sober_kmeans_ws_bytes_screened(N, d, K)) or int(lib.sober_kmeans_ws_bytes(N, d, K))
        a, listed = run(full); b, _ = run(0)
        bad = int((a != b).sum().item())
        tot_pts += 10 * N; tot_bad += bad
        rows.append("%s %-8s listed %.5f  labels that differ after 10 iterations: %d" % (shape, kind, listed / (10.0 * N), bad))
print("\n".join(rows))
print("TOTAL point-iterations %d, differing labels %d" % (tot_pts, tot_bad))
