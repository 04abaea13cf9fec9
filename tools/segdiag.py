#!/usr/bin/env python3
"""Diagnostics of the segment kernel on the bench workload (GPU box):
    python tools/segdiag.py [n_streams]
  * how many streams inflate_segments_kernel finishes itself (the rest goes to the slow kernels)
  * time of the segment kernel alone (FDH_FLAG_FIRST_ONLY) vs the whole pipeline
  * with FDH_LIB pointing at a -DFDH_DEBUG_TILES build: cycles per phase of a stream
    (window walk / count / check / scan / write / tail), from clock64() stamps of lane 0."""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import fdeflate_amd as fd  # noqa: E402
from fdeflate_amd import _lib, synth  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
model = sys.argv[2] if len(sys.argv) > 2 else "D"
L = 65536
dev = "cuda"
raw = synth.gen_batch_torch(0, n, L, model=model, device=dev)
r_off = torch.arange(n + 1, dtype=torch.int64, device=dev) * L
bound = (fd.ultrafast_bound(L) + 15) & ~15
t_off = torch.arange(n + 1, dtype=torch.int64, device=dev) * bound
tmp = torch.zeros(n * bound, dtype=torch.uint8, device=dev)
clen = fd.deflate_ultrafast_batch(raw.view(-1), r_off, tmp, t_off)
del tmp
# the bench's layout: packed, every stream 16-B aligned
t_off = torch.zeros(n + 1, dtype=torch.int64, device=dev)
t_off[1:] = torch.cumsum((clen.to(torch.int64) + 15) & ~15, 0)
comp = torch.zeros(int(t_off[-1]), dtype=torch.uint8, device=dev)
fd.deflate_ultrafast_batch(raw.view(-1), r_off, comp, t_off)
out = torch.empty(n * L, dtype=torch.uint8, device=dev)
ol = torch.empty(n, dtype=torch.int32, device=dev)
st = torch.empty(n, dtype=torch.int32, device=dev)
ad = torch.empty(n, dtype=torch.int32, device=dev)


def timed(flags, reps=5):
    fd.inflate_batch(comp, t_off, out, r_off, ol, st, ad, flags=flags)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fd.inflate_batch(comp, t_off, out, r_off, ol, st, ad, flags=flags)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


t_first = timed(64)
pending = int((st == -1).sum())
t_all = timed(0)
ok = int((st == 0).sum())
print("streams %d model %s mean compressed %.0f B" % (n, model, float(clen.sum()) / n))
print("segment kernel alone: %.3f ms, finished %d of %d streams itself; whole pipeline %.3f ms (ok %d), equal output: %s"
      % (t_first, n - pending, n, t_all, ok, bool(torch.equal(out, raw.view(-1)))))
print("=> scaled to 65536 streams: %.2f ms" % (t_all * 65536 / n))
Lc = _lib.lib()
if hasattr(Lc, "fdh_debug_read_handed"):
    hb = np.zeros(65537, dtype=np.uint32)
    Lc.fdh_debug_read_handed(hb.ctypes.data_as(C.c_void_p), 1)
    fd.inflate_batch(comp, t_off, out, r_off, ol, st, ad, flags=64)
    torch.cuda.synchronize()
    Lc.fdh_debug_read_handed(hb.ctypes.data_as(C.c_void_p), 1)
    k = min(n, 65536)
    print("stream hand-out: min %d max %d times per stream (first %d streams), hand-outs under a partial EXEC: %d"
          % (hb[:k].min(), hb[:k].max(), k, hb[65536]))
if hasattr(Lc, "fdh_debug_read_segtime"):
    fd.inflate_batch(comp, t_off, out, r_off, ol, st, ad, flags=64)
    torch.cuda.synchronize()
    buf = np.zeros(4096 * 8, dtype=np.uint32)
    Lc.fdh_debug_read_segtime(buf.ctypes.data_as(C.c_void_p))
    t = buf.reshape(4096, 8).astype(np.int64)
    d = (t[:, 1:7] - t[:, 0:6]) & 0xFFFFFFFF
    names = ["window walk", "count", "check", "scan+setup", "write", "tail"]
    kinds = np.arange(4096) % 16
    for label, sel in (("noisy streams", (kinds != 15) & (kinds != 7)), ("half-zero", kinds == 7), ("all-zero", kinds == 15)):
        m = d[sel][: min(n, 4096)].mean(axis=0)
        print("%-14s total %8.0f cycles: " % (label, m.sum()) + ", ".join("%s %.0f" % (a, b) for a, b in zip(names, m)))
    dbg = np.zeros(64 * 16, dtype=np.uint32)
    Lc.fdh_debug_read_seg(dbg.ctypes.data_as(C.c_void_p))
    dbg = dbg.reshape(64, 16)
    dbg2 = np.zeros(16 * 16, dtype=np.uint32)
    Lc.fdh_debug_read_seg2(dbg2.ctypes.data_as(C.c_void_p))
    dbg2 = dbg2.reshape(16, 16)
    d2 = dbg2.reshape(-1)
    for k in range(4):
        r = d2[8 * k: 8 * k + 8]
        print("write pass stream %d: start %d, loop %d = events %d + drain %d + groups %d (%d) + general %d (%d) + other %d cycles"
              % (k, r[6], r[7], r[0], r[1], r[2], r[4], r[3], r[5], int(r[7]) - int(r[0]) - int(r[1]) - int(r[2]) - int(r[3])))
    c = dbg[:4, 0]
    print("count scan: general steps %s, fast groups %s, fast lanes per group %s" % (c & 0xFF, (c >> 8) & 0xFF, (c >> 16) / np.maximum((c >> 8) & 0xFF, 1)))
    print("iterations (stream 0..3): count-scan %s window %s head %s recount %s rounds %s write %s"
          % (dbg[:4, 0], dbg[:4, 6], dbg[:4, 1], dbg[:4, 2], dbg[:4, 3], dbg[:4, 4]))
