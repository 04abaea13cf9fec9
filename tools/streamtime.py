#!/usr/bin/env python3
"""Wall time, attempts and decoded bytes of the streaming Decompressor for a large stream (GPU box):
python tools/streamtime.py           (FDH_STREAM_NO_RESUME=1 python tools/streamtime.py  for the A/B)
A 3 MB buffer as a zlib level-6 stream and in the ultra-fast format; (a) the whole input at once, drained through a
16 KiB window with 32 KiB of history (the png crate's pattern); (b) the input in 32 KiB pieces into a large buffer."""
import os, sys, time, zlib
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import fdeflate_amd as fd

r = np.random.default_rng(77)
raw = bytes((np.cumsum(r.integers(-3, 4, size=3_000_000)) & 0xFF).astype(np.uint8))
uf = fd.compress_to_vec_ultra_fast(raw)
for name, comp in (("zlib-6", zlib.compress(raw, 6)), ("ultra-fast", uf)):
    d = fd.Decompressor(); d.read(comp[:100], bytearray(1000), 0)   # warm-up (tables, first launches)
    t0 = time.perf_counter()
    d = fd.Decompressor()
    got = bytearray(); buf = bytearray(32768 + 16384); pos = 0; k = 0; calls = 0
    while not d.is_done():
        calls += 1
        c, p = d.read(comp[k:], buf, pos)   # (what is not consumed is offered again)
        k += c
        got += buf[pos:pos + p]; pos += p
        if pos > 32768:
            buf[:32768] = buf[pos - 32768:pos]; pos = 32768
    ta = time.perf_counter() - t0
    assert bytes(got) == raw
    print("%-10s window 16 KiB : %7.1f ms  %4d calls %3d attempts  decoded %5.2f x the stream, device memory %d KiB" % (name, ta * 1e3, calls, d.attempts(), d.decoded_bytes() / len(raw), d.device_bytes() >> 10))
    t0 = time.perf_counter()
    d = fd.Decompressor()
    buf = bytearray(len(raw) + 64); pos = 0; calls = 0
    k = 0
    while k < len(comp):
        c, p = d.read(comp[k:k + 32768], buf, pos); pos += p; k += c; calls += 1
    while not d.is_done():
        c, p = d.read(b"", buf, pos); pos += p; calls += 1
    tb = time.perf_counter() - t0
    assert bytes(buf[:pos]) == raw
    print("%-10s input 32 KiB   : %7.1f ms  %4d calls %3d attempts  decoded %5.2f x the stream, device memory %d KiB" % (name, tb * 1e3, calls, d.attempts(), d.decoded_bytes() / len(raw), d.device_bytes() >> 10))
