"""Time sober_level_reduce_tani alone at level 0 of configuration 5 (250k x 2048-bit fingerprints, 700 rows, 200 sets).
Library from SOBER_HIP_LIB: the timing-experiment builds (-DTANI_X_NODSREAD / NOMFMA / NOQUOT / NOEXPAND) give wrong
sums on purpose -- they show which part of the kernel its time is made of."""
import ctypes as C, os, sys, json
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sober_amd import _native as nat
dev = torch.device("cuda")
n, n_rows, S, dt = 250000, 700, 200, 32
g = torch.Generator().manual_seed(0)
cand = torch.randint(-2**62, 2**62, (n, dt), generator=g, dtype=torch.int64).to(dev)
rows = torch.randint(-2**62, 2**62, (n_rows, dt), generator=g, dtype=torch.int64).to(dev)
cn = torch.full((n,), 1000.0, dtype=torch.float64, device=dev); rn = torch.full((n_rows,), 1000.0, dtype=torch.float64, device=dev)
idx = torch.arange(n, dtype=torch.int32, device=dev)
mu = torch.rand(n, generator=g, dtype=torch.float64).to(dev)
P = nat.level_chunks(n_rows, 0, n, S)
partG = torch.empty(64 * n_rows * S, dtype=torch.float64, device=dev); partTot = torch.empty(64 * S, dtype=torch.float64, device=dev)
lib = nat.load(); st = torch.cuda.current_stream().cuda_stream
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
us = e0.elapsed_time(e1) * 100
print(json.dumps({"lib": os.path.basename(os.environ.get("SOBER_HIP_LIB", "head")), "chunks": P, "us": round(us, 1),
                  "int8_TOPS": round(n * n_rows * 2 * 2048 / us / 1e6, 1)}))
