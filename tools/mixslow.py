#!/usr/bin/env python3
"""Which streams of the mixed batch (bench.py, BASELINE config 5) are slow: every distinct stream of the pool
decoded as a batch of 256 copies, timed on its own: python tools/mixslow.py [flags]   (GPU box)"""
import os, sys, random, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import bench
import fdeflate_amd as fd
from fdeflate_amd import synth

flags = int(sys.argv[1], 0) if len(sys.argv) > 1 else 0
only = [int(x) for x in sys.argv[2].split(",")] if len(sys.argv) > 2 else None   # entries of the pool to look at (e.g. under rocprofv3)
dev = torch.device("cuda", 0)
n, L = 64, 65536
raw = synth.gen_batch_torch(0, n, L, device=dev)
r_off = torch.arange(n + 1, dtype=torch.int64, device=dev) * L
comp, c_off, clen = bench.encode_ultrafast(raw, r_off, dev)
pool, nref = bench.mix_pool(raw, comp, c_off, clen, random.Random(2024))
res = []
for k, (c, r, ok, what) in enumerate(pool):
    if only is not None and k not in only:
        continue
    m = 256
    cap = len(r) if r is not None else 65536
    buf = torch.from_numpy(np.frombuffer(c * m + bytes(16), dtype=np.uint8).copy()).to(dev)
    off = torch.arange(m + 1, dtype=torch.int64, device=dev) * len(c)
    ooff = torch.arange(m + 1, dtype=torch.int64, device=dev) * cap
    out = torch.empty(m * cap + 16, dtype=torch.uint8, device=dev)
    ln = torch.empty(m, dtype=torch.int32, device=dev); st = torch.empty(m, dtype=torch.int32, device=dev); ad = torch.empty(m, dtype=torch.int32, device=dev)
    fd.inflate_batch(buf, off, out, ooff, ln, st, ad, flags=flags)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        fd.inflate_batch(buf, off, out, ooff, ln, st, ad, flags=flags)
    torch.cuda.synchronize()
    res.append(((time.perf_counter() - t0) / 3 * 1e3, k, len(c), int(st[0]), what))
res.sort(reverse=True)
for ms, k, lc, st, what in res[:25]:
    print("%8.3f ms  #%3d  %6d B  status %2d  %s" % (ms, k, lc, st, what))
