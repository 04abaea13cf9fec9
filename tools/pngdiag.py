#!/usr/bin/env python3
"""Throughput of the PNG scanline kernels on the bench's buffers (64 KiB of filtered rows per image):
python tools/pngdiag.py [n_images] [bpp]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import fdeflate_amd as fd  # noqa: E402
from fdeflate_amd import synth  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
bpp = int(sys.argv[2]) if len(sys.argv) > 2 else 3
L = 65536
dev = "cuda"
raw = synth.gen_batch_torch(0, n, L, device=dev)      # rows of synth.ROW_BYTES + 1 (filter type first)
row = synth.ROW_BYTES - 1
rows = L // (row + 1)
flen = rows * (row + 1)
filt = raw[:, :flen].contiguous()
f_off = torch.arange(n + 1, dtype=torch.int64, device=dev) * flen
p_off = torch.arange(n + 1, dtype=torch.int64, device=dev) * (rows * row)
pix = torch.empty(n * rows * row, dtype=torch.uint8, device=dev)
types = filt.view(n, rows, row + 1)[:, :, 0].contiguous().view(-1)
t_off = torch.arange(n + 1, dtype=torch.int64, device=dev) * rows
st = fd.png_unfilter_batch(filt.view(-1), f_off, pix, p_off, row, bpp)
print("row_bytes %d, %d rows per image, filter types %s, status ok %d / %d" %
      (row, rows, torch.bincount(types.to(torch.int64), minlength=5).tolist(), int((st == 0).sum()), n))
back = torch.empty_like(filt.view(-1))
for name, call, nbytes in (
        ("unfilter", lambda: fd.png_unfilter_batch(filt.view(-1), f_off, pix, p_off, row, bpp), n * flen),
        ("filter", lambda: fd.png_filter_batch(pix, p_off, types, t_off, back, f_off, row, bpp), n * flen)):
    call()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    reps = 5
    for _ in range(reps):
        call()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    print("%-9s %8.2f ms  %8.1f GB/s (filtered bytes)" % (name, dt * 1e3, nbytes / dt / 1e9))
print("filter(unfilter(x)) == x:", bool(torch.equal(back, filt.view(-1))))
