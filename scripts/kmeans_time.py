"""Time the Nystrom subsample (KMeans, 10 Lloyd iterations, K = 500) at the pool sizes of configurations 2 and 4: the
library's default route, the E step screened on the BF16 matrix cores (forced where the shape allows) and, with a workspace
one byte short of what that needs, the FP64-only E step of round 3; the share of points the exact pass had to decide."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sober_amd
from sober_amd import _native as nat
dev = torch.device("cuda")
lib = nat.load()
sizes = [tuple(int(v) for v in t.split(":")) for t in os.environ.get("KM_SIZES", "100000:10,1000000:20").split(",")]
K = int(os.environ.get("KM_K", "500"))
for N, d in sizes:
    g = torch.Generator().manual_seed(0)
    X = torch.rand(N, d, generator=g, dtype=torch.float64).to(dev)
    full = int(lib.sober_kmeans_ws_bytes_screened(N, d, K)) or int(lib.sober_kmeans_ws_bytes(N, d, K))
    off = int(lib.sober_kmeans_stat_offset(N, d, K))
    res = {}
    dflt = int(lib.sober_kmeans_ws_bytes(N, d, K))            # (what sober_amd.KMeans asks for: screened from the size on where it pays)
    for name, nbytes in (("default", dflt), ("screened", full), ("fp64 only", full - 1 if off >= 0 else full)):
        c = torch.empty(K, d, dtype=torch.float64, device=dev); cl = torch.empty(N, dtype=torch.int32, device=dev)
        ws = torch.zeros(full, dtype=torch.uint8, device=dev)
        run = lambda: nat._check(lib.sober_kmeans_lloyd(X.data_ptr(), N, d, K, 10, c.data_ptr(), cl.data_ptr(), ws.data_ptr(),
                                                        nbytes, nat._stream(X)), "kmeans")
        run(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5): run()
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / 5 * 1e3
        listed = int(ws[off:off + 4].view(torch.int32).item()) if (off >= 0 and nbytes == full) else 0
        res[name] = (cl.clone(), c.clone())
        print("N %d d %d K %d %-10s %.3f ms per KMeans, listed %.4f of the points, centroid checksum %.12f"
              % (N, d, K, name, ms, listed / (10.0 * N), float(c.nan_to_num().sum())))
    print("   labels equal:", bool(torch.equal(res["screened"][0], res["fp64 only"][0])) and bool(torch.equal(res["default"][0], res["fp64 only"][0])))
t0 = time.perf_counter()
for N, d in sizes[:1]:
    X = torch.rand(N, d, dtype=torch.float64, device=dev)
    for _ in range(3): sober_amd.KMeans(X, K)
    torch.cuda.synchronize()
