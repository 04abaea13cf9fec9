#!/usr/bin/env python3
"""Where the general encoder's parser spends its time, per stream (debug library):
python tools/gentimers.py [n_streams]"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("FDH_LIB", os.path.join(ROOT, "fdeflate_amd", "libfdeflate_hip_debug.so"))
import numpy as np  # noqa: E402
import torch  # noqa: E402

import fdeflate_amd as fd  # noqa: E402
from fdeflate_amd import _lib, synth  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
L = 65536
raw = synth.gen_batch_torch(0, n, L, device="cuda")
r_off = torch.arange(n + 1, dtype=torch.int64, device="cuda") * L
bound = (fd.compress_bound(L) + 15) & ~15
off = torch.arange(n + 1, dtype=torch.int64, device="cuda") * bound
out = torch.empty(n * bound, dtype=torch.uint8, device="cuda")
lib = _lib.lib()
names = ["scan (advance_to_match)", "  of which rle_match", "  of which match_length8", "inserts behind a match", "look at the next position", "", "", "whole stream"]
for mode, label in ((fd.MODE_LEVEL1, "level 1"), (fd.MODE_RLE, "rle")):
    fd.deflate_general_batch(raw.view(-1), r_off, out, off, mode)
    torch.cuda.synchronize()
    t = np.zeros((8, 32768), dtype=np.uint64)
    assert lib.fdh_debug_gen_timers(t.ctypes.data_as(C.c_void_p)) == 0
    t = t[:, :min(n, 32768)].astype(np.float64)
    print("== %s: cycles per stream, mean / max over %d streams" % (label, t.shape[1]))
    for k, nm in enumerate(names):
        if nm:
            print("%-28s %12.0f %12.0f" % (nm, t[k].mean(), t[k].max()))
    w = np.zeros((12, 32768), dtype=np.uint32)
    assert lib.fdh_debug_gen_write_timers(w.ctypes.data_as(C.c_void_p)) == 0
    w = w[:, :min(n, 32768)].astype(np.float64)
    print("== %s, block writer: clocks per stream (mean), share" % label)
    for k, nm in enumerate(["walk 1 (frequencies)", "prepare", "depths (one lane)", "limit / codes", "header", "walk 2 (symbols)", "block set-up", "whole stream", "  merges, literal/length tree", "  merges, distance tree", "  merges, code-length tree", ""]):
        if not nm:
            continue
        print("%-28s %12.0f %6.1f %%" % (nm, w[k].mean(), 100 * w[k].mean() / w[7].mean()))
    heavy = np.argsort(t[7])[-3:]
    print("heaviest streams:", heavy.tolist(), (t[:, heavy] / 1e3).astype(int).T.tolist())
