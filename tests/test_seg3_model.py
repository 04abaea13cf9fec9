"""The counting rules of the landing decoder (fdeflate_amd/csrc/inflate_seg3.h) as an executable CPU
model (tests/seg3_model.py): guessed chains, exact landings on the right neighbour's start, run chains
that end their interval.  Checked against the raw bytes the streams were made from (the oracle's
ultra-fast encoder, reference src/compress/ultrafast.rs:94-181) and against zlib."""
import json
import os
import zlib

import numpy as np

import oracle_binding as ob
import seg3_model as m
from fdeflate_amd import synth

HERE = os.path.dirname(os.path.abspath(__file__))
G = json.load(open(os.path.join(HERE, "golden", "constants.json")))
LENGTHS = (list(G["HUFFMAN_LENGTHS"]) + [0, 0])[:288]
CANON_BITS = 53 * 8 + 5  # reference src/compress/ultrafast.rs:82-88: 53 bytes and 5 bits of prefix


def _check(raw, repeat=m.REPEAT):
    comp = ob.compress_ultra_fast(raw)
    assert zlib.decompress(comp) == raw
    # (as the kernel: no segment shorter than 1 024 bits -- a short stream uses fewer lanes)
    nseg = min(64, max(1, (len(comp) * 8 - CANON_BITS) // 1024))
    total, lanes, stats = m.plan(comp, LENGTHS, CANON_BITS, nseg=nseg, repeat=repeat)
    assert total == len(raw), stats
    # every lane lands on its neighbour's start; the intervals of a lane are at most METER steps long and
    # their byte counts add up
    for l, (x0, end, cnt, ck) in enumerate(lanes):
        if l + 1 < len(lanes):
            assert end == lanes[l + 1][0]
        assert ck[0] == (x0, 0) and ck[-1] == (end, cnt)
        assert all(ck[k][0] <= ck[k + 1][0] and ck[k][1] <= ck[k + 1][1] for k in range(len(ck) - 1))
    return stats


def test_bench_streams_land():
    for i in (1, 2, 7, 9):  # noisy rows; every other row zero (long run chains)
        stats = _check(synth.gen_stream_np(i).tobytes())
        assert stats["fail"] == 0


def test_long_chains_of_the_lean_writer():
    """Round 6: a stream the lean writer takes merges up to 64 run tokens into a chain (an all-zero 64 KiB buffer is
    four chains, not 32): the same bytes, the same landings, fewer intervals."""
    for i in (7, 15):  # every other row zero; all zero
        raw = synth.gen_stream_np(i).tobytes()
        a = _check(raw)
        b = _check(raw, repeat=m.REPEAT_LEAN)
        assert b["fail"] == 0 and b["chains"] <= a["chains"]
    z = _check(bytes(65536), repeat=m.REPEAT_LEAN)
    assert z["chains"] <= 6, z


def test_other_models_and_lengths_land():
    rng = np.random.default_rng(5)
    for model, length in (("M", 40000), ("L", 50000), ("U", 20000)):
        _check(synth.gen_stream_np(3, length, model, png_rows=False).tobytes())
    # runs of every length from 8 up to a few hundred between noisy stretches
    parts = []
    for r in list(range(8, 40)) + [257, 258, 259, 260, 516, 517, 600, 1031]:
        parts.append(rng.integers(1, 256, size=37, dtype=np.uint8).tobytes())
        parts.append(bytes(r))
    _check(b"".join(parts) * 3)
