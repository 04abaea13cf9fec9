#!/usr/bin/env python3
"""Throughput of the encoders on the bench workload: python tools/encdiag.py [n_streams]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import fdeflate_amd as fd  # noqa: E402
from fdeflate_amd import synth  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
L = 65536
dev = "cuda"
raw = synth.gen_batch_torch(0, n, L, device=dev)
r_off = torch.arange(n + 1, dtype=torch.int64, device=dev) * L
for name, bound, call in (
        ("ultra-fast", (fd.ultrafast_bound(L) + 15) & ~15, lambda o, off, ln: fd.deflate_ultrafast_batch(raw.view(-1), r_off, o, off, ln)),
        ("stored", (fd.stored_size(L) + 15) & ~15, lambda o, off, ln: fd.deflate_stored_batch(raw.view(-1), r_off, o, off, ln)),
        ("rle", (fd.compress_bound(L) + 15) & ~15, lambda o, off, ln: fd.deflate_general_batch(raw.view(-1), r_off, o, off, fd.MODE_RLE, ln)),
        ("level 1", (fd.compress_bound(L) + 15) & ~15, lambda o, off, ln: fd.deflate_general_batch(raw.view(-1), r_off, o, off, fd.MODE_LEVEL1, ln))):
    off = torch.arange(n + 1, dtype=torch.int64, device=dev) * bound
    out = torch.empty(n * bound, dtype=torch.uint8, device=dev)
    ln = torch.empty(n, dtype=torch.int32, device=dev)
    call(out, off, ln)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    reps = 3
    for _ in range(reps):
        call(out, off, ln)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    print("%-10s %8.2f ms  %8.1f GB/s in, ratio %.3f" % (name, dt * 1e3, n * L / dt / 1e9, float(ln.to(torch.int64).sum()) / (n * L)))
    del out
