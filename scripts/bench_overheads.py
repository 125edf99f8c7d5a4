"""What bench.py's own per-step bookkeeping costs on the host (not the product's work): torch.manual_seed, the warnings
context, the reset of the weights."""
import time, warnings, torch
dev = torch.device("cuda:0")
torch.zeros(1, device=dev)
mu0 = torch.rand(100000, dtype=torch.float64, device=dev); mu = mu0.clone()
def t(f, n=2000):
    for _ in range(50): f()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): f()
    return (time.perf_counter() - t0) / n * 1e6
print("torch.manual_seed                      %.1f us" % t(lambda: torch.manual_seed(1234)))
print("torch.default_generator.manual_seed    %.1f us" % t(lambda: torch.default_generator.manual_seed(1234)))
def w():
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
print("warnings.catch_warnings + simplefilter %.1f us" % t(w))
print("mu.copy_(mu0) (host side)              %.1f us" % t(lambda: mu.copy_(mu0)))
