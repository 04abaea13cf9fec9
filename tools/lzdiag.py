#!/usr/bin/env python3
"""The LZ-window kernel on its own (FDH_FLAG_LZ_ONLY): which streams it finishes, and that what it
finishes is right.  python tools/lzdiag.py"""
import os
import random
import sys
import zlib

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402

import gpu_harness as gh  # noqa: E402
import streams  # noqa: E402
from fdeflate_amd import synth  # noqa: E402

LZ_ONLY = 0x2000
rnd = random.Random(5)
cases = []
for name, comp, raw in streams.valid_streams():
    cases.append((name, comp, raw))
for lvl in (1, 4, 6, 9):
    for sid, ln in ((0, 65536), (7, 65536), (15, 65536), (3, 20000), (4, 300), (9, 200000)):
        raw = synth.gen_stream_np(sid, ln).tobytes()
        cases.append(("zlib%d_s%d_%d" % (lvl, sid, ln), zlib.compress(raw, lvl), raw))
for strat, sname in ((zlib.Z_FIXED, "fixed"), (zlib.Z_RLE, "rle"), (zlib.Z_HUFFMAN_ONLY, "huff"), (zlib.Z_FILTERED, "filt")):
    raw = synth.gen_stream_np(11, 50000).tobytes()
    c = zlib.compressobj(6, zlib.DEFLATED, 15, 9, strat)
    cases.append(("strat_" + sname, c.compress(raw) + c.flush(), raw))
text = (b"the quick brown fox jumps over the lazy dog " * 400) + bytes(rnd.randrange(256) for _ in range(3000))
cases.append(("text", zlib.compress(text * 5, 6), text * 5))
raw = b"ab" * 5000 + b"xyz" * 3000 + b"q" * 70000
cases.append(("overlap", zlib.compress(raw, 9), raw))
raw = bytes(rnd.randrange(256) for _ in range(200)) + bytes(30000) + bytes(rnd.randrange(4) for _ in range(40000))
cases.append(("far", zlib.compress(raw, 6), raw))
raw = bytes(1 << 20)
cases.append(("zeros1M", zlib.compress(raw, 6), raw))
raw = bytes(rnd.randrange(3) for _ in range(1 << 20))
cases.append(("rand3_1M", zlib.compress(raw, 6), raw))

names = [c[0] for c in cases]
blobs = [c[1] for c in cases]
for slack in (0, 100):
    caps = [len(c[2]) + slack for c in cases]
    st, ln, ad, outs, guards_ok = gh.gpu_inflate(blobs, caps, flags=LZ_ONLY)
    took = 0
    bad = []
    for i, (name, comp, raw) in enumerate(cases):
        if st[i] == 0:
            took += 1
            if ln[i] != len(raw) or outs[i][:len(raw)].tobytes() != raw or ad[i] != (zlib.adler32(raw) & 0xFFFFFFFF):
                got = outs[i][:len(raw)]
                ref = np.frombuffer(raw, dtype=np.uint8)
                nd = np.nonzero(got[:min(len(got), len(ref))] != ref[:min(len(got), len(ref))])[0]
                bad.append((name, int(ln[i]), len(raw), int(nd[0]) if len(nd) else -1, len(nd)))
        elif st[i] != 0xFFFFFFFF:
            bad.append((name, "status", int(st[i])))
    left = [names[i] for i in range(len(cases)) if st[i] == 0xFFFFFFFF]
    print("slack %d: LZ kernel finished %d of %d, guards %s" % (slack, took, len(cases), guards_ok))
    print("  left:", left)
    print("  WRONG:", bad)
    # whole pipeline
    st, ln, ad, outs, guards_ok = gh.gpu_inflate(blobs, caps, flags=0)
    wrong = [names[i] for i, c in enumerate(cases) if st[i] != 0 or outs[i][:len(c[2])].tobytes() != c[2]]
    print("  whole pipeline wrong:", wrong, "guards", guards_ok)
