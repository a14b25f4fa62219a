#!/usr/bin/env python3
"""Do two HIP streams overlap on this box?  Stream A: a 1 GiB device copy (~400 us); stream B: a 300 us sleep kernel."""
import time, torch
a = torch.empty(1 << 28, dtype=torch.float32, device="cuda"); b = torch.empty_like(a)
sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
def run(concurrent):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        with torch.cuda.stream(sa):
            b.copy_(a, non_blocking=True)
        with torch.cuda.stream(sb if concurrent else sa):
            torch.cuda._sleep(300 * 2100)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / 20 * 1e6
for _ in range(2):
    print("same stream %.1f us, two streams %.1f us per iteration" % (run(False), run(True)))
