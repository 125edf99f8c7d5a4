"""Host LAPACK SVD of the Caratheodory step's (m x 2m) matrix at batch 200 (the route batches > 100 take) as a
function of the torch thread count on this box."""
import time, torch
torch.manual_seed(0)
for m in (100, 200):
    A = torch.randn(m, 2 * m, dtype=torch.float64); A[0] = 1.0
    for th in (1, 2, 4, 8, 16):
        torch.set_num_threads(th)
        for _ in range(3): torch.linalg.svd(A)
        t0 = time.perf_counter()
        for _ in range(10): torch.linalg.svd(A)
        print("m=%d threads=%2d  %.2f ms" % (m, th, (time.perf_counter() - t0) / 10 * 1e3))
