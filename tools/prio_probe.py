#!/usr/bin/env python3
"""Developer probe: do HIP stream priorities decide who gets the CUs when two streams have kernels pending?
Two streams each run N elementwise passes over their own 256 MiB tensor; reports when each stream finished (ms after the start)
for (normal, normal) and (normal, high). usage: python tools/prio_probe.py"""
import torch

dev = torch.device("cuda", 0)
print("priority range:", torch.cuda.Stream.priority_range() if hasattr(torch.cuda.Stream, "priority_range") else "n/a")


def trial(pa, pb, n=60):
    sa, sb = torch.cuda.Stream(dev, priority=pa), torch.cuda.Stream(dev, priority=pb)
    xa, xb = torch.zeros(64 << 20, device=dev), torch.zeros(64 << 20, device=dev)
    torch.cuda.synchronize()
    e0, ea, eb = (torch.cuda.Event(enable_timing=True) for _ in range(3))
    e0.record()
    sa.wait_event(e0); sb.wait_event(e0)
    for _ in range(n):
        with torch.cuda.stream(sa):
            xa.add_(1.0)
        with torch.cuda.stream(sb):
            xb.add_(1.0)
    ea.record(sa); eb.record(sb)
    torch.cuda.synchronize()
    return e0.elapsed_time(ea), e0.elapsed_time(eb)


trial(0, 0); trial(0, -1)
for pa, pb in ((0, 0), (0, -1), (-1, 0), (0, 0), (-1, -1), (0, -1), (-1, 0)):
    a, b = trial(pa, pb)
    print(f"priorities ({pa:2d}, {pb:2d}): stream a done at {a:7.2f} ms, stream b at {b:7.2f} ms")
