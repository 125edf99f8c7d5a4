"""Does a CU-masked side stream run beside the main stream's small dependent kernels on this box?  (round 6: level 0's set sums
beside the Nystrom chain)  hipExtStreamCreateWithCUMask through ctypes, torch.cuda.ExternalStream around the handle."""
import ctypes, time, sys
import torch
hip = ctypes.CDLL("libamdhip64.so")
dev = torch.device("cuda:0")
torch.cuda.init(); torch.zeros(1, device=dev)
n_cu = torch.cuda.get_device_properties(dev).multi_processor_count
print("CUs", n_cu)

def masked_stream(keep_free):
    """A stream that may use every CU except `keep_free` of them (taken from the high bits of every 32-CU word)."""
    words = (n_cu + 31) // 32
    per = keep_free // words
    mask = (ctypes.c_uint32 * words)(*[(0xFFFFFFFF >> per) for _ in range(words)])
    s = ctypes.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(ctypes.byref(s), ctypes.c_uint32(words), mask)
    print("hipExtStreamCreateWithCUMask rc", rc, "mask", [hex(m) for m in mask])
    return torch.cuda.ExternalStream(s.value, device=dev) if rc == 0 else None

big_a = torch.rand(8192, 8192, dtype=torch.float64, device=dev)
small = [torch.rand(64, 64, dtype=torch.float64, device=dev) for _ in range(2)]

def chain(n=60):
    x = small[0]
    for _ in range(n):
        x = (x @ small[1]) * 0.01
    return x

def big():
    return big_a @ big_a

for name, side in (("plain side stream", torch.cuda.Stream(device=dev)), ("masked side stream (32 CUs kept free)", masked_stream(32))):
    if side is None:
        continue
    for _ in range(2):
        big(); chain()
    torch.cuda.synchronize()
    t0 = time.perf_counter(); big(); torch.cuda.synchronize(); t_big = time.perf_counter() - t0
    t0 = time.perf_counter(); chain(); torch.cuda.synchronize(); t_chain = time.perf_counter() - t0
    with torch.cuda.stream(side):
        big()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    with torch.cuda.stream(side):
        big()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); chain(); e1.record()
    torch.cuda.synchronize()
    t_both = time.perf_counter() - t0
    print(f"{name}: big alone {t_big*1e3:.2f} ms, chain alone {t_chain*1e3:.2f} ms, both {t_both*1e3:.2f} ms, the chain beside the big one {e0.elapsed_time(e1):.2f} ms")
