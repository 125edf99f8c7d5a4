"""When every pivot of k_car_pivot_stream is published (diagnostic build -DSP_TSTAMPS: make -C sober_amd/csrc
BUILD=build_sps EXTRA='-DSOBER_DIAG_BUILD -DSP_TSTAMPS' OUT=build_sps/libsober_hip_sps.so), on the reference's level-0 input of the
matern_medium golden (N = 200, m = 100): microseconds between consecutive publishes, in-block pivots against the first pivot
of a block (the hand-over from the wave before)."""
import os as _os; _os.environ.setdefault("SOBER_ALLOW_DIAG_LIB", "1")   # (a stamped library is a diagnostic build)
import ctypes as C, numpy as np, torch, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sober_amd import _native as nat
dev = torch.device("cuda:0")
z = np.load(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "recomb_matern_medium.npz"))
X, mu = np.ascontiguousarray(z["L0_X_tmp"]), z["L0_tot_weights"]
N, n = X.shape
Xd, mud = torch.from_numpy(X).to(dev), torch.from_numpy(mu).to(dev)
kr = torch.empty(N, dtype=torch.int32, device=dev); ws = torch.empty(N, dtype=torch.float64, device=dev)
nk = torch.empty(1, dtype=torch.int32, device=dev); mo = torch.empty(N, dtype=torch.float64, device=dev)
lib = nat.load()
lib.sober_debug_sp_stamps.restype = C.c_int
acc = []
for it in range(8):
    nat.car_device(Xd, mud, kr, ws, nk, mo)
    torch.cuda.synchronize()
    buf = (C.c_ulonglong * 260)()
    assert lib.sober_debug_sp_stamps(buf) == 0
    t = np.array(buf[:], dtype=np.int64)
    K = N - (n + 1)
    if it >= 3:
        acc.append((t[:K] - t[256]) / 100.0)
        tail = (t[257] - t[K - 1]) / 100.0
a = np.mean(acc, 0)
d = np.diff(np.concatenate([[0.0], a]))
first = np.arange(len(d)) % 7 == 0
print("pivots", len(d), "first publish after the barrier: %.2f us; last publish at %.2f us; tail (weights out) %.2f us" % (d[0], a[-1], tail))
print("in-block pivots : mean %.3f us (min %.3f max %.3f)" % (d[~first].mean(), d[~first].min(), d[~first].max()))
print("hand-over pivots: mean %.3f us (min %.3f max %.3f)  [%d of them]" % (d[first][1:].mean(), d[first][1:].min(), d[first][1:].max(), first.sum() - 1))
print("by position in the block:", [round(float(d[(np.arange(len(d)) % 7 == j) & (np.arange(len(d)) >= 7)].mean()), 3) for j in range(7)])
print("per block (us):", [round(float(d[b * 7:(b + 1) * 7].sum()), 2) for b in range((len(d) + 6) // 7)])
seg = (C.c_ulonglong * 128)()
if hasattr(lib, "sober_debug_sp_segments") and lib.sober_debug_sp_segments(seg) == 0:
    sg = np.array(seg[:], dtype=np.float64).reshape(16, 8)[:14]           # the 14 full blocks
    per = sg.sum(0) / (14 * 7) / 100.0
    print("inside the ratio test (screened form): rcp, keys, lane minimum %.3f | wave minimum %.3f | ballots, winner's lane, read-outs %.3f | the winner's two divisions %.3f" % (per[5], per[6], per[7], per[1]))
    print("produce step by segment, us per pivot (each includes one stamp's own cost): elimination behind the previous pivot + column select %.3f | "
          "ratio test %.3f | publish %.3f | weights %.3f | (block end) %.3f" % (per[0], per[1], per[2], per[3], per[4] * 7))
rt = (C.c_ulonglong * 8)()
if hasattr(lib, "sober_debug_car_rt") and lib.sober_debug_car_rt(rt) == 0:
    r = np.array(rt[:], dtype=np.int64)
    t = np.array(buf[:], dtype=np.int64)
    o = r[0]
    f = lambda x: (x - o) / 100.0
    print("timeline of the step's two launches (us, producer's entry = 0): last reflector computed %.2f | published, producer out %.2f | "
          "last consumer (Phi) done %.2f | pivot kernel in %.2f | its barrier %.2f | first publish %.2f | last publish %.2f | weights out %.2f"
          % (f(r[1]), f(r[2]), f(r[3]), f(t[258]), f(t[256]), f(t[0]), f(t[K - 1]), f(t[257])))
