#!/usr/bin/env python3
"""Per-phase clocks of the LZ-window kernel on zlib level-6 streams of the bench data (library built with
-DFDH_LZ_DEBUG: tools/build_lzdebug.sh):  python tools/lztime.py [n_streams] [level]"""
import ctypes as C
import os
import sys
import zlib
from concurrent.futures import ThreadPoolExecutor

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("FDH_LIB", os.path.join(ROOT, "fdeflate_amd", "libfdeflate_hip_lzdebug.so"))
import numpy as np  # noqa: E402
import torch  # noqa: E402

import fdeflate_amd as fd  # noqa: E402
from fdeflate_amd import _lib, synth  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
level = int(sys.argv[2]) if len(sys.argv) > 2 else 6
L = 65536
dev = "cuda"
raw = synth.gen_batch_torch(0, n, L, device=dev)
h = raw.cpu().numpy()
with ThreadPoolExecutor(16) as pool:
    blobs = list(pool.map(lambda i: zlib.compress(h[i].tobytes(), level), range(n), chunksize=64))
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
Lc = _lib.lib()
g = (C.c_ulonglong * 32)()
fd.inflate_batch(comp, c_off, out, r_off, ol, st, ad)
torch.cuda.synchronize()
Lc.fdh_debug_read_lzstat(g, 1)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
fd.inflate_batch(comp, c_off, out, r_off, ol, st, ad)
e1.record()
torch.cuda.synchronize()
Lc.fdh_debug_read_lzstat(g, 1)
g = [int(x) for x in g]
ns = max(g[31], 1)
names = ["header", "stage", "pass1", "fixups", "offsets", "pass2", "resolve", "flush", "trailer"]
tot = sum(g[:9]) + sum(g[13:18])
print("%d streams finished by the LZ kernel, %.2f ms (instrumented), ok %d, equal %s" % (g[31], e0.elapsed_time(e1), int((st == 0).sum()), bool(torch.equal(out, raw.view(-1)))))
print("cycles per stream: %.0f" % (tot / ns))
for k, nm in enumerate(names):
    print("  %-8s %9.0f  %5.1f %%" % (nm, g[k] / ns, 100.0 * g[k] / max(tot, 1)))
print("  spans %.1f  fix-up rounds %.1f  match batches %.0f  ordered rounds %.0f (%.2f per batch)  - %.0f" %
      (g[9] / ns, g[10] / ns, g[11] / ns, g[12] / ns, g[12] / max(g[11], 1), g[13] / ns))
print("  walk iterations per stream: pass 1 + fix-ups %.0f (slow steps %.0f), pass 2 %.0f (slow steps %.0f)" % (g[21] / ns, g[23] / ns, g[20] / ns, g[22] / ns))
print("  cycles per iteration: pass 1 + fix-ups %.0f, pass 2 %.0f" % ((g[2] + g[3]) / max(g[21], 1), g[5] / max(g[20], 1)))
print("  header parts: code-length code %.0f, lengths chain %.0f, ranking %.0f, table fill %.0f, second level %.0f" % tuple(g[k] / ns for k in (13, 14, 15, 16, 17)))
