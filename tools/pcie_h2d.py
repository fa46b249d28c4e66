#!/usr/bin/env python3
"""What the link gives: pinned host -> device copy rate for the size of a 1024-frame crop upload (192 MB) and for whole
frames (944 MB), beside the host-fed entry point's figure.   python3 tools/pcie_h2d.py"""
import time

import torch

dev = torch.device('cuda', 0)
for mb in (192, 944):
    n = mb * 1000 * 1000
    src = torch.empty(n, dtype=torch.uint8).pin_memory()
    dst = torch.empty(n, dtype=torch.uint8, device=dev)
    for _ in range(3):
        dst.copy_(src, non_blocking=True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    reps = 10
    for _ in range(reps):
        dst.copy_(src, non_blocking=True)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    print('pinned H2D %4d MB: %.3f ms = %.1f GB/s  -> at most %.0f K frames/s of 187 500-byte crops' % (mb, dt * 1e3, n / dt / 1e9, n / dt / 187500 / 1e3))
