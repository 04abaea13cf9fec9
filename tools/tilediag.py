#!/usr/bin/env python3
"""Where the general (dynamic-block) decoder spends its time on zlib level-6 streams of the bench
data (debug library built by tools/build_all.sh, -DFDH_DEBUG_TILES):
    python tools/tilediag.py [n_streams] [level]
Prints the throughput of the product library and the per-phase clocks of the tile decoder."""
import ctypes as C
import os
import sys
import zlib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
if os.environ.get("FDH_TILE_DEBUG", "1") == "1":
    os.environ.setdefault("FDH_LIB", os.path.join(ROOT, "fdeflate_amd", "libfdeflate_hip_debug.so"))
import numpy as np  # noqa: E402
import torch  # noqa: E402

import fdeflate_amd as fd  # noqa: E402
from fdeflate_amd import _lib, synth  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
level = int(sys.argv[2]) if len(sys.argv) > 2 else 6
L = 65536
dev = "cuda"
raw = synth.gen_batch_torch(0, n, L, device=dev)
rows = raw.view(n, L).cpu().numpy()
blobs = [zlib.compress(rows[i].tobytes(), level) for i in range(n)]
clen = np.array([len(b) for b in blobs], dtype=np.int64)
off_h = np.zeros(n + 1, dtype=np.int64)
off_h[1:] = np.cumsum((clen + 15) & ~15)
buf = np.zeros(int(off_h[-1]), dtype=np.uint8)
for i, b in enumerate(blobs):
    buf[off_h[i]:off_h[i] + len(b)] = np.frombuffer(b, dtype=np.uint8)
comp = torch.from_numpy(buf).to(dev)
c_off = torch.from_numpy(off_h).to(dev)
r_off = torch.arange(n + 1, dtype=torch.int64, device=dev) * L
out = torch.empty(n * L, dtype=torch.uint8, device=dev)
ol = torch.empty(n, dtype=torch.int32, device=dev)
st = torch.empty(n, dtype=torch.int32, device=dev)
ad = torch.empty(n, dtype=torch.int32, device=dev)
flags = int(os.environ.get("FDH_TILE_FLAGS", "0"), 0)


def run():
    fd.inflate_batch(comp, c_off, out, r_off, ol, st, ad, flags=flags)


run()
torch.cuda.synchronize()
ok = bool(torch.equal(out, raw.view(-1))) and bool((st == 0).all())
best = 1e9
for _ in range(3):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    run()
    e1.record()
    torch.cuda.synchronize()
    best = min(best, e0.elapsed_time(e1))
print("streams %d level %d mean compressed %.0f B: %.3f ms, %.1f GB/s decompressed, all right: %s"
      % (n, level, clen.mean(), best, n * L / best / 1e6, ok))
lib = _lib.lib()
if hasattr(lib, "fdh_debug_read_gstat"):
    g = np.zeros(24, dtype=np.uint64)
    lib.fdh_debug_read_gstat(g.ctypes.data_as(C.c_void_p), 1)
    run()
    torch.cuda.synchronize()
    lib.fdh_debug_read_gstat(g.ctypes.data_as(C.c_void_p), 1)
    g = g.astype(np.float64)
    tiles = max(g[0], 1)
    print("per stream: %.1f tiles (%.0f stream bits each, %.1f short ones), %.1f serial tokens, %.1f headers, %.0f matches"
          % (g[0] / n, g[1] / tiles, g[5] / n, g[3] / n, g[9] / n, g[15] / n))
    print("cycles per stream: tiles %.0f, serial %.0f, headers %.0f" % (g[2] / n, g[7] / n, g[8] / n))
    names = {10: "pass 1 (guessed chains)", 11: "synchronisation (%.1f iterations per tile)" % (g[6] / tiles),
             12: "counts + prefix sums", 13: "literals, match list", 16: "matches with sources older than the ring", 14: "the other matches (rounds, one by one)"}
    print("matches per tile: %.1f, copied in %.1f rounds (%.1f per round) and %.1f one by one"
          % (g[15] / tiles, g[19] / tiles, g[21] / max(g[19], 1), g[20] / tiles))
    print("cycles per tile:")
    for k in (10, 11, 12, 13, 16, 14):
        print("   %-44s %8.0f" % (names[k], g[k] / tiles))
