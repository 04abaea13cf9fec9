#!/usr/bin/env python3
"""Times fdh_inflate_batch on the bench workload WITHOUT checking the output: for elimination experiments, where a
library built with a piece of the kernel compiled out (wrong bytes on purpose) shows what that piece costs.
    FDH_LIB=ab/lib_x.so python tools/xtime.py [n_streams] [steps]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402
import fdeflate_amd as fd  # noqa: E402
from fdeflate_amd import synth  # noqa: E402

n, L = (int(sys.argv[1]) if len(sys.argv) > 1 else 65536), 65536
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 40
dev = torch.device("cuda", 0)
raw = synth.gen_batch_torch(0, n, L, device=dev)
if os.environ.get("XNORUN"):  # no aligned 8-byte chunk ends in a zero: the encoder then never starts a run (no run tokens at all)
    v = raw.view(n, L // 8, 8)
    v[:, :, 7] = torch.where(v[:, :, 7] == 0, torch.ones_like(v[:, :, 7]), v[:, :, 7])
r_off = torch.arange(n + 1, dtype=torch.int64, device=dev) * L
comp, c_off, clen = bench.encode_ultrafast(raw, r_off, dev)
out = torch.empty(n * L, dtype=torch.uint8, device=dev)
ol = torch.empty(n, dtype=torch.int32, device=dev)
st = torch.empty(n, dtype=torch.int32, device=dev)
ad = torch.empty(n, dtype=torch.int32, device=dev)
flags = int(os.environ.get("XFLAGS", "0"), 0)
for _ in range(10):
    fd.inflate_batch(comp, c_off, out, r_off, ol, st, ad, flags=flags)
torch.cuda.synchronize()
ev = [torch.cuda.Event(enable_timing=True) for _ in range(steps + 1)]
ev[0].record()
for k in range(steps):
    fd.inflate_batch(comp, c_off, out, r_off, ol, st, ad, flags=flags)
    ev[k + 1].record()
torch.cuda.synchronize()
ts = sorted(ev[k].elapsed_time(ev[k + 1]) for k in range(steps))
ok = int((st == 0).sum())
print("%-40s median %.4f ms  min %.4f  (status Ok: %d of %d)" % (os.environ.get("FDH_LIB", "product"), ts[len(ts) // 2], ts[0], ok, n))
