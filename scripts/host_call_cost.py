"""Host-side cost of one native call (Python -> ctypes -> hipLaunchKernel) and of its parts, next to a torch op."""
import time, torch, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sober_amd import _native as nat
dev = torch.device("cuda:0")
A = torch.rand(64, 64, dtype=torch.float64, device=dev); B = torch.empty_like(A); flag = torch.zeros(1, dtype=torch.int32, device=dev)
def per(fn, n=3000):
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n): fn()
    dt = (time.perf_counter() - t) / n * 1e6
    torch.cuda.synchronize(); return dt
print("torch.cuda.current_stream(dev).cuda_stream  %.2f us" % per(lambda: torch.cuda.current_stream(dev).cuda_stream))
print("A.data_ptr()                                %.2f us" % per(lambda: A.data_ptr()))
print("nat.abs_sym (64 x 64)                       %.2f us" % per(lambda: nat.abs_sym(A, B, flag)))
lib = nat.load(); st = torch.cuda.current_stream(dev).cuda_stream; pa, pb, pf = A.data_ptr(), B.data_ptr(), flag.data_ptr()
print("raw ctypes sober_abs_sym                    %.2f us" % per(lambda: lib.sober_abs_sym(pa, 64, 64, pb, 64, pf, st)))
print("torch A.add_(1.0)                           %.2f us" % per(lambda: A.add_(1.0)))
print("torch.empty(64,64)                          %.2f us" % per(lambda: torch.empty(64, 64, dtype=torch.float64, device=dev)))
C = torch.empty(64, 64, dtype=torch.float64, device=dev)
print("nat.dgemm 64^3                              %.2f us" % per(lambda: nat.dgemm(A, B, C)))
raw = getattr(torch._C, "_cuda_getCurrentRawStream", None)
if raw is not None:
    print("torch._C._cuda_getCurrentRawStream(0)           %.2f us" % per(lambda: raw(0)), "same handle:", raw(0) == torch.cuda.current_stream(dev).cuda_stream)
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        print("   inside a stream context: same handle:", raw(0) == side.cuda_stream)
