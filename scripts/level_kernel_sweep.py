"""Time sober_level_reduce_mfma alone at one level's shape:
   python scripts/level_kernel_sweep.py [d=10] [n=100000] [rows=700] [S=200] -- P ...
P = number of partial sums the launch is given; 0 = what sober_level_parts_mfma says (the only value the wave-autonomous
build accepts; the workgroup-staged build, -DSOBER_LM_BLOCK, takes any).  Library from SOBER_HIP_LIB (scripts/level_kernel_sweep.sh)."""
import os, subprocess, sys, json

def child(d, n, n_rows, S, P):
    import torch
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from sober_amd import _native as nat
    dev = torch.device("cuda")
    g = torch.Generator().manual_seed(0)
    X = torch.rand(n, d, generator=g, dtype=torch.float64).to(dev)
    R = torch.rand(n_rows, d, generator=g, dtype=torch.float64).to(dev)
    ls = torch.full((1,), 0.7 * d ** 0.5, dtype=torch.float64, device=dev)
    center = X.mean(0)
    da = nat.load().sober_aug_dim(d)
    cand = torch.empty(n, da, dtype=torch.float64, device=dev)
    rows = torch.empty(n_rows, da, dtype=torch.float64, device=dev)
    nat.augment_points(R, ls, center, 0, rows)
    nat.augment_points(X, ls, center, 1, cand)
    idx = torch.arange(n, dtype=torch.int32, device=dev)
    mu = torch.rand(n, generator=g, dtype=torch.float64).to(dev)
    partG = torch.empty(64 * n_rows * S, dtype=torch.float64, device=dev)
    partTot = torch.empty(64 * S, dtype=torch.float64, device=dev)
    E = n // S
    if P == 0: P = nat.level_parts_mfma(n_rows, 0, n, S)
    def run():
        nat.level_reduce_mfma(int(os.environ.get("LK_KIND", "0")), rows, cand, da, idx, 0, 0, n, S, mu, None, 1.3, P, partG, S, 0, partTot, E * S)   # LK_KIND=1: Matern-5/2
    for _ in range(3): run()
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    reps = 20
    ev[0].record()
    for _ in range(reps): run()
    ev[1].record(); torch.cuda.synchronize()
    us = ev[0].elapsed_time(ev[1]) * 1e3 / reps
    flop = n * n_rows * (2 * d + 2 + 28)
    print(json.dumps({"P": P, "n": n, "us": round(us, 2), "TF": round(flop / us / 1e6, 2),
                      "frac": round(flop / us / 1e6 / 78.6, 3), "chk": float(partG[: n_rows * S].sum())}), flush=True)

if __name__ == "__main__":
    if sys.argv[1] == "--child":
        d, n, n_rows, S, P = (int(x) for x in sys.argv[2:7]); child(d, n, n_rows, S, P); sys.exit(0)
    args = sys.argv[1:]
    k = args.index("--")
    kv = dict(a.split("=") for a in args[:k])
    d, n, n_rows, S = int(kv.get("d", 10)), int(kv.get("n", 100000)), int(kv.get("rows", 700)), int(kv.get("S", 200))
    for P in args[k + 1:]:
        subprocess.run([sys.executable, __file__, "--child", str(d), str(n), str(n_rows), str(S), P])
