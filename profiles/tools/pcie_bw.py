#!/usr/bin/env python3
"""pcie_bw.py -- host <-> device copy rates of the box (pinned memory, 1 GiB transfers, both directions alone and together)"""
import time
import torch

n = 1 << 30
h_in = torch.empty(n, dtype=torch.uint8, pin_memory=True)
h_out = torch.empty(n, dtype=torch.uint8, pin_memory=True)
d_a = torch.empty(n, dtype=torch.uint8, device="cuda")
d_b = torch.empty(n, dtype=torch.uint8, device="cuda")
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()


def timed(fn, reps=5):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps


def h2d():
    with torch.cuda.stream(s1):
        d_a.copy_(h_in, non_blocking=True)


def d2h():
    with torch.cuda.stream(s2):
        h_out.copy_(d_b, non_blocking=True)


def both():
    h2d(); d2h()


print(f"H2D {n / timed(h2d) / 1e9:.1f} GB/s   D2H {n / timed(d2h) / 1e9:.1f} GB/s   both at once: {n / timed(both) / 1e9:.1f} GB/s each")
