"""The CPU model of the LZ-window decoder (tests/lz_model.py) reproduces zlib's output."""
import os
import random
import sys
import zlib

import numpy as np
import pytest

sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
from fdeflate_amd.synth import gen_stream_np  # noqa: E402
from tests.lz_model import Bail, LzModel  # noqa: E402


def _cases():
    rnd = random.Random(7)
    yield "bench0", zlib.compress(gen_stream_np(0).tobytes(), 6)
    yield "halfzero", zlib.compress(gen_stream_np(7).tobytes(), 6)
    yield "zeros", zlib.compress(gen_stream_np(15).tobytes(), 6)
    yield "level9", zlib.compress(gen_stream_np(3, 20000).tobytes(), 9)
    yield "level1", zlib.compress(gen_stream_np(4, 30000).tobytes(), 1)
    c = zlib.compressobj(6, zlib.DEFLATED, 15, 9, zlib.Z_FIXED)
    yield "fixed", c.compress(gen_stream_np(5, 9000).tobytes()) + c.flush()
    c = zlib.compressobj(6, zlib.DEFLATED, 15, 9, zlib.Z_RLE)
    yield "rle", c.compress(gen_stream_np(7, 40000).tobytes()) + c.flush()
    c = zlib.compressobj(6, zlib.DEFLATED, 15, 9, zlib.Z_HUFFMAN_ONLY)
    yield "huffman_only", c.compress(gen_stream_np(6, 5000).tobytes()) + c.flush()
    text = (b"the quick brown fox jumps over the lazy dog " * 400) + bytes(rnd.randrange(256) for _ in range(3000))
    yield "text", zlib.compress(text * 3, 6)
    yield "tiny", zlib.compress(b"abcabcabcabc", 6)
    yield "one", zlib.compress(b"a", 6)
    yield "overlap", zlib.compress(b"ab" * 5000 + b"xyz" * 3000 + b"q" * 70000, 9)
    yield "far", zlib.compress(bytes(rnd.randrange(256) for _ in range(200)) + bytes(30000) +
                                 bytes(rnd.randrange(4) for _ in range(40000)), 6)


_CASES = list(_cases())


@pytest.mark.parametrize("name,data", _CASES, ids=[c[0] for c in _CASES])
def test_model_matches_zlib(name, data):
    want = zlib.decompress(data)
    for R, W, cap in ((544, 512, 8192), (96, 64, 2048)):
        try:
            got = LzModel(data, R=R, W=W, img_cap=cap).run()
        except Bail as ex:
            pytest.fail("%s: model gave up: %s" % (name, ex))
        assert got == want, name
