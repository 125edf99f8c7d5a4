"""Time the Nystrom subsample (KMeans, 10 Lloyd iterations, K = 500) at the pool sizes of configurations 2 and 4."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sober_amd
dev = torch.device("cuda")
sizes = [tuple(int(v) for v in t.split(":")) for t in os.environ.get("KM_SIZES", "100000:10,1000000:20").split(",")]
for N, d in sizes:
    g = torch.Generator().manual_seed(0)
    X = torch.rand(N, d, generator=g, dtype=torch.float64).to(dev)
    sober_amd.KMeans(X, 500); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3): cl, c = sober_amd.KMeans(X, 500)
    torch.cuda.synchronize()
    print("N %d d %d: %.2f ms per KMeans, centroid checksum %.12f" % (N, d, (time.perf_counter() - t0) / 3 * 1e3, float(c.nan_to_num().sum())))
