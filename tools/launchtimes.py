#!/usr/bin/env python3
"""Duration of every single fdh_inflate_batch call in a row of them (bench workload), HIP events
around each call: python tools/launchtimes.py [n_calls] [n_streams]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import fdeflate_amd as fd  # noqa: E402
from fdeflate_amd import synth  # noqa: E402

calls = int(sys.argv[1]) if len(sys.argv) > 1 else 60
n = int(sys.argv[2]) if len(sys.argv) > 2 else 65536
L = 65536
dev = "cuda"
raw = synth.gen_batch_torch(0, n, L, model="D", device=dev)
r_off = torch.arange(n + 1, dtype=torch.int64, device=dev) * L
bound = (fd.ultrafast_bound(L) + 15) & ~15
t_off = torch.arange(n + 1, dtype=torch.int64, device=dev) * bound
tmp = torch.zeros(n * bound, dtype=torch.uint8, device=dev)
clen = fd.deflate_ultrafast_batch(raw.view(-1), r_off, tmp, t_off)
del tmp
t_off = torch.zeros(n + 1, dtype=torch.int64, device=dev)
t_off[1:] = torch.cumsum((clen.to(torch.int64) + 15) & ~15, 0)
comp = torch.zeros(int(t_off[-1]), dtype=torch.uint8, device=dev)
fd.deflate_ultrafast_batch(raw.view(-1), r_off, comp, t_off)
out = torch.empty(n * L, dtype=torch.uint8, device=dev)
ol = torch.empty(n, dtype=torch.int32, device=dev)
st = torch.empty(n, dtype=torch.int32, device=dev)
ad = torch.empty(n, dtype=torch.int32, device=dev)
torch.cuda.synchronize()
ev = [torch.cuda.Event(enable_timing=True) for _ in range(calls + 1)]
ev[0].record()
for i in range(calls):
    fd.inflate_batch(comp, t_off, out, r_off, ol, st, ad)
    ev[i + 1].record()
torch.cuda.synchronize()
ms = [ev[i].elapsed_time(ev[i + 1]) for i in range(calls)]
print("per call, ms:", " ".join("%.2f" % x for x in ms))
s = sorted(ms)
print("min %.3f median %.3f mean %.3f max %.3f" % (s[0], s[len(s) // 2], sum(ms) / len(ms), s[-1]))
