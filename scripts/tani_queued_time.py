"""the queued form of the fingerprint level kernel (level size read from device memory, launch sized from the cap) against
the plain form with the chunk count the device finds: same time expected"""
import ctypes as C, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sober_amd import _native as nat
dev = torch.device("cuda")
lib = nat.load(); st = torch.cuda.current_stream().cuda_stream
g = torch.Generator().manual_seed(0)
n_rows, S, dt = 700, 200, 32
N = 250000
cand = torch.randint(-2**62, 2**62, (N, dt), generator=g, dtype=torch.int64).to(dev)
rows = torch.randint(-2**62, 2**62, (n_rows, dt), generator=g, dtype=torch.int64).to(dev)
cn = torch.full((N,), 1000.0, dtype=torch.float64, device=dev); rn = torch.full((n_rows,), 1000.0, dtype=torch.float64, device=dev)
mu = torch.rand(N, generator=g, dtype=torch.float64).to(dev)
partG = torch.zeros(64 * n_rows * S, dtype=torch.float64, device=dev); partTot = torch.zeros(64 * S, dtype=torch.float64, device=dev)
lib.sober_level_reduce_tani_queued.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int64, C.c_int, C.c_int, C.c_int,
                                               C.c_void_p, C.c_void_p, C.c_double, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
def timed(fn):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): fn()
    e1.record(); torch.cuda.synchronize()
    return round(e0.elapsed_time(e1) * 100, 1)
for n, scattered in ((125000, False), (125000, True), (62500, True)):
    idx = (torch.randperm(N, generator=g)[:n].sort().values if scattered else torch.arange(n)).to(torch.int32).to(dev)
    dR = torch.tensor([n], dtype=torch.int64, device=dev)
    P = int(lib.sober_level_chunks_tani(n_rows, 0, n, S)); cap = int(lib.sober_level_chunks_tani_cap(n_rows, (n + S - 1) // S, S))
    plain = timed(lambda: lib.sober_level_reduce_tani(rows.data_ptr(), rn.data_ptr(), n_rows, cand.data_ptr(), cn.data_ptr(), dt, idx.data_ptr(),
                                                      0, n, S, mu.data_ptr(), None, 1.3, P, partG.data_ptr(), S, 0, partTot.data_ptr(), (n // S) * S, st))
    queued = timed(lambda: lib.sober_level_reduce_tani_queued(rows.data_ptr(), rn.data_ptr(), n_rows, cand.data_ptr(), cn.data_ptr(), dt, idx.data_ptr(),
                                                             n, S, S, 0, mu.data_ptr(), None, 1.3, cap, partG.data_ptr(), S, partTot.data_ptr(), dR.data_ptr(), st))
    print("n", n, "scattered list" if scattered else "contiguous list", "chunks", P, "cap", cap, "plain us", plain, "queued us", queued)
