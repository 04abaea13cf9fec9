#!/usr/bin/env python3
"""General (zlib) decode path on level-6 streams of the bench data: python tools/gendiag.py [n] [flags...]"""
import os
import sys
import zlib
from concurrent.futures import ThreadPoolExecutor

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import fdeflate_amd as fd  # noqa: E402
from fdeflate_amd import synth  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
flag_sets = [int(x, 0) for x in sys.argv[2:]] or [0]
L = 65536
dev = "cuda"
raw = synth.gen_batch_torch(0, n, L, device=dev)
h = raw.cpu().numpy()
with ThreadPoolExecutor(16) as pool:
    blobs = list(pool.map(lambda i: zlib.compress(h[i].tobytes(), 6), range(n), chunksize=64))
clen = np.array([len(b) for b in blobs], dtype=np.int64)
off = np.zeros(n + 1, dtype=np.int64)
off[1:] = np.cumsum((clen + 15) & ~15)
buf = np.zeros(int(off[-1]), dtype=np.uint8)
for i, b in enumerate(blobs):
    buf[off[i]:off[i] + len(b)] = np.frombuffer(b, dtype=np.uint8)
comp, c_off = torch.from_numpy(buf).to(dev), torch.from_numpy(off).to(dev)
r_off = torch.arange(n + 1, dtype=torch.int64, device=dev) * L
out = torch.empty(n * L, dtype=torch.uint8, device=dev)
ol = torch.empty(n, dtype=torch.int32, device=dev)
st = torch.empty(n, dtype=torch.int32, device=dev)
ad = torch.empty(n, dtype=torch.int32, device=dev)
for flags in flag_sets:
    out.zero_()
    fd.inflate_batch(comp, c_off, out, r_off, ol, st, ad, flags=flags)
    torch.cuda.synchronize()
    ok = int((st == 0).sum())
    same = bool(torch.equal(out, raw.view(-1)))
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    reps = 3
    for _ in range(reps):
        fd.inflate_batch(comp, c_off, out, r_off, ol, st, ad, flags=flags)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    from fdeflate_amd import _lib
    import ctypes as C
    Lc = _lib.lib()
    if hasattr(Lc, "fdh_debug_read_gstat"):
        g = (C.c_ulonglong * 24)()
        Lc.fdh_debug_read_gstat(g, 1)
        fd.inflate_batch(comp, c_off, out, r_off, ol, st, ad, flags=flags)
        torch.cuda.synchronize()
        Lc.fdh_debug_read_gstat(g, 1)
        g = [int(x) for x in g]
        per = lambda v: v / n
        print("  per stream: tiles %.1f (%.0f bits, %.0f cycles)  spans %.1f (%.0f bits, %.0f cycles: pass1 %.0f check %.0f pass2 %.0f matches %.0f [%d per stream] tail %.0f)  headers %.1f (%.0f cycles)"
              % (per(g[0]), per(g[1]), per(g[2]), per(g[3]), per(g[4]), per(g[7]), per(g[10]), per(g[11]), per(g[12]), per(g[13]), per(g[15]), per(g[14]), per(g[9]), per(g[8])))
    print("flags %#x: %.2f ms for %d streams = %.1f GB/s decompressed (ok %d, bytes equal %s)" % (flags, ms, n, n * L / ms / 1e6, ok, same))
