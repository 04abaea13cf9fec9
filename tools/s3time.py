#!/usr/bin/env python3
"""Per-phase clocks of the landing decoder (inflate_seg3.h) on the bench workload (GPU box):
    tools/build_s3debug.sh && FDH_LIB=fdeflate_amd/libfdeflate_hip_debug.so python tools/s3time.py [n_streams]
Prints, for the streams the kernel finished among the first 4 096, the mean clock64 ticks per phase."""
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
Lc = _lib.lib()
base = int(sys.argv[2]) if len(sys.argv) > 2 else 0  # sample the 4 096 streams from this one on (a multiple of 16)
assert Lc.fdh_debug_s3base(C.c_uint32(base)) == 0
for _ in range(3):
    fd.inflate_batch(comp, c_off, out, r_off, ol, st, ad, flags=0)
torch.cuda.synchronize()
buf = np.zeros(4096 * 16 + 16 + 4096 * 16 + 4096 * 8, dtype=np.uint32)
assert Lc.fdh_debug_s3time(buf.ctypes.data_as(C.c_void_p)) == 0
t = buf[:4096 * 16].reshape(4096, 16).astype(np.int64)
names = ["set-up + prefix", "ring fill", "guess", "periods", "groups + pairs", "singles", "plan", "write"]
def report(title, sel):
    if not sel:
        return
    d = np.array([[(t[i, k + 1] - t[i, k]) & 0xFFFFFFFF for k in range(8)] for i in sel], dtype=np.float64)
    med = np.median(d, axis=0)
    tot = med.sum()
    print("%s streams: %d, total %.0f ticks per stream (medians)" % (title, len(sel), tot))
    for k, nm in enumerate(names):
        print("  %-16s %9.0f  %5.1f %%" % (nm, med[k], 100 * med[k] / tot))
    w = buf[4096 * 16 + 16:4096 * 16 + 16 + 4096 * 16].reshape(4096, 16).astype(np.float64)
    wn = ["round top", "stage a + c / request", "lane set-up (wait for input)", "group", "chains", "checks", "flush", "carry"]
    wm = np.median(w[sel][:, 8:16], axis=0)
    if wm.sum() > 0:
        print("writing pass (seg2_write clocks), %.0f ticks:" % wm.sum())
        for k, nm in enumerate(wn):
            print("  %-30s %9.0f  %5.1f %%" % (nm, wm[k], 100 * wm[k] / wm.sum()))
    lw = buf[4096 * 16 + 16 + 4096 * 16:].reshape(4096, 8).astype(np.float64)
    ln = ["fetch (owner search, requests)", "wait for input + lane set-up", "group", "chains", "settle / fits / request", "flush", "tail / carry", "round top"]
    lm = np.median(lw[sel], axis=0)
    if lm.sum() > 0:
        print("lean writing pass (seg3_write clocks), %.0f ticks:" % lm.sum())
        for k, nm in enumerate(ln):
            print("  %-30s %9.0f  %5.1f %%" % (nm, lm[k], 100 * lm[k] / lm.sum()))
    print("  periods per stream %.1f, with a run chain %.1f" % (np.mean([t[i, 10] for i in sel]), np.mean([t[i, 11] for i in sel])))
    print("  writing rounds per stream %.1f, intervals %.1f (%.1f per round)" % (np.mean([t[i, 12] for i in sel]), np.mean([t[i, 13] for i in sel]),
          np.mean([t[i, 13] for i in sel]) / max(1e-9, np.mean([t[i, 12] for i in sel]))))
    print("  rounds cut short by the input image %.1f, by the output image %.1f" % (np.mean([t[i, 14] for i in sel]), np.mean([t[i, 15] for i in sel])))


done = [i for i in range(min(n, 4096)) if t[i, 8] != 0]
report("noisy", [i for i in done if i % 16 not in (7, 15)])
report("half-zero", [i for i in done if i % 16 == 7])
report("all-zero", [i for i in done if i % 16 == 15])
if os.environ.get("S3DUMP"):
    np.save(os.environ["S3DUMP"], t)
print("stat:", buf[4096 * 16:4096 * 16 + 4])
if os.environ.get("S3RAW"):
    for i in done[:6]:
        print(i, [int(x) for x in t[i, :12]])
