#!/usr/bin/env python3
"""Timeline of the landing decoder's wavefronts on the bench workload (GPU box, instrumented build):
    tools/build_s3debug.sh && FDH_LIB=fdeflate_amd/libfdeflate_hip_debug.so python tools/s3seq.py [n_streams]
Per wavefront: when it started, which streams it took and when each was done (s_memrealtime, 100 MHz).  Prints the
spread of the starts, the duration of a stream by its place in the wavefront's sequence and by kind, and how long the
wavefronts sit idle at the kernel's end."""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

import bench  # noqa: E402
import fdeflate_amd as fd  # noqa: E402
from fdeflate_amd import _lib, synth  # noqa: E402

n, L = (int(sys.argv[1]) if len(sys.argv) > 1 else 65536), 65536
dev = torch.device("cuda", 0)
raw = synth.gen_batch_torch(0, n, L, device=dev)
r_off = torch.arange(n + 1, dtype=torch.int64, device=dev) * L
comp, c_off, clen = bench.encode_ultrafast(raw, r_off, dev)
out = torch.empty(n * L, dtype=torch.uint8, device=dev)
ol = torch.empty(n, dtype=torch.int32, device=dev)
st = torch.empty(n, dtype=torch.int32, device=dev)
ad = torch.empty(n, dtype=torch.int32, device=dev)
for _ in range(3):
    fd.inflate_batch(comp, c_off, out, r_off, ol, st, ad, flags=0)
torch.cuda.synchronize()
Lc = _lib.lib()
buf = np.zeros(4096 * 64, dtype=np.uint32)
assert Lc.fdh_debug_s3seq(buf.ctypes.data_as(C.c_void_p)) == 0
w = buf.reshape(4096, 64).astype(np.int64)
cl = clen.cpu().numpy() if hasattr(clen, "cpu") else np.asarray(clen)
t0 = w[:, 63].min()
start = (w[:, 63] - t0) / 100.0  # us
cnt = w[:, 0]
ends = np.array([(w[i, 2 * cnt[i]] - t0) / 100.0 if cnt[i] else start[i] for i in range(4096)])
print("wavefronts: start spread %.1f us (p50 %.1f, p99 %.1f); streams per wavefront min %d mean %.1f max %d" %
      (start.max(), np.median(start), np.percentile(start, 99), cnt.min(), cnt.mean(), cnt.max()))
staged = (w[:, 62] - w[:, 63]) / 100.0
print("tables staged after %.1f us (p50; p99 %.1f)" % (np.median(staged), np.percentile(staged, 99)))
print("kernel: last wavefront done at %.1f us; mean done %.1f; idle at the end: mean %.1f us, p10 %.1f, p90 %.1f" %
      (ends.max(), ends.mean(), (ends.max() - ends).mean(), np.percentile(ends.max() - ends, 10), np.percentile(ends.max() - ends, 90)))
mean_len = cl.mean()
by_place = {}
by_kind = {"noisy": [], "half": [], "zero": []}
for i in range(4096):
    prev = w[i, 63]
    for k in range(min(cnt[i], 30)):
        sid, t = w[i, 1 + 2 * k], w[i, 2 + 2 * k]
        d = ((t - prev) & 0xFFFFFFFF) / 100.0
        prev = t
        c = cl[sid]
        kind = "zero" if c < 2000 else ("half" if c < 0.75 * mean_len * 1.1 else "noisy")
        by_kind[kind].append(d)
        if kind == "noisy":
            by_place.setdefault(k, []).append(d)
for kind, v in by_kind.items():
    if v:
        print("  %-6s %6d streams: mean %.1f us, p10 %.1f, p50 %.1f, p90 %.1f" % (kind, len(v), np.mean(v), np.percentile(v, 10), np.median(v), np.percentile(v, 90)))
print("noisy streams by place in the wavefront's sequence (us, mean):")
print("  " + " ".join("%d:%.0f" % (k, np.mean(v)) for k, v in sorted(by_place.items())))
print("sum of stream times per wavefront: mean %.1f us; kernel %.1f us" % (np.mean([ends[i] - start[i] for i in range(4096)]), ends.max()))
