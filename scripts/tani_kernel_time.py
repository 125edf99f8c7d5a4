"""sober_level_reduce_tani alone: its sums (checksum + a dump for a bit-for-bit comparison between two libraries) at three
fingerprint lengths with real popcounts, ragged sizes, zero weights and a weight multiplier, then its time at level 0 of
configuration 5 (250k x 2048-bit fingerprints, 700 rows, 200 sets).  Library from SOBER_HIP_LIB.
   python scripts/tani_kernel_time.py [dump.npz]"""
import ctypes as C, os, sys, json
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sober_amd import _native as nat
dev = torch.device("cuda")
lib = nat.load(); st = torch.cuda.current_stream().cuda_stream
g = torch.Generator().manual_seed(0)
dump = {}
for (n, n_rows, S, dt, pos0) in [(5003, 137, 20, 8, 0), (20011, 700, 200, 16, 0), (9000, 300, 64, 32, 1280), (250000, 700, 200, 32, 0)]:
    cand = (torch.randint(-2**62, 2**62, (n, dt), generator=g, dtype=torch.int64) & torch.randint(-2**62, 2**62, (n, dt), generator=g, dtype=torch.int64)).to(dev)
    rows = torch.randint(-2**62, 2**62, (n_rows, dt), generator=g, dtype=torch.int64).to(dev)
    pc = lambda t: sum(((t >> s) & 1) for s in range(64)).sum(1).to(torch.float64)
    cn, rn = pc(cand), pc(rows)
    idx = torch.randperm(n, generator=g).to(torch.int32).to(dev)
    mu = torch.rand(n, generator=g, dtype=torch.float64).to(dev); mu[::7] = 0.0
    wm = torch.rand(n, generator=g, dtype=torch.float64).to(dev)
    count = n - pos0
    P = int(lib.sober_level_chunks_tani(n_rows, pos0, count, S)) if hasattr(lib, "sober_level_chunks_tani") else nat.level_chunks(n_rows, pos0, count, S)
    partG = torch.zeros(64 * n_rows * S, dtype=torch.float64, device=dev); partTot = torch.zeros(64 * S, dtype=torch.float64, device=dev)
    def run():
        rc = lib.sober_level_reduce_tani(rows.data_ptr(), rn.data_ptr(), n_rows, cand.data_ptr(), cn.data_ptr(), dt, idx.data_ptr(),
                                         pos0, count, S, mu.data_ptr(), wm.data_ptr(), 1.3, P, partG.data_ptr(), S, 0, partTot.data_ptr(),
                                         ((pos0 + count) // S) * S, st)
        assert rc == 0, rc
    for _ in range(3): run()
    torch.cuda.synchronize()
    key = "n%d_r%d_S%d_dt%d" % (n, n_rows, S, dt)
    dump[key + "_G"] = partG[:P * n_rows * S].cpu().numpy(); dump[key + "_T"] = partTot[:P * S].cpu().numpy()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): run()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 100
    print(json.dumps({"lib": os.path.basename(os.environ.get("SOBER_HIP_LIB", "head")), "case": key, "chunks": P, "us": round(us, 1),
                      "P_bit_pair_ops_per_s": round(count * n_rows * 2 * 64 * dt / us / 1e9, 3), "checksum": float(dump[key + "_G"].sum())}))
if len(sys.argv) > 1:
    np.savez(sys.argv[1], **dump)
