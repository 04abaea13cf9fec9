"""The speculative tile-decode ALGORITHM (tests/tile_model.py) against zlib on the CPU."""
import zlib

import numpy as np

import oracle_binding as ob
import streams
import tile_model as tm
from fdeflate_amd import synth


def _decode_ultrafast_with_tiles(comp, golden_constants):
    lit, dist = tm.build_tables(golden_constants["HUFFMAN_LENGTHS"], [1])
    model = tm.TileModel(comp, lit, dist)
    P = 53 * 8 + 5
    while True:
        used, eob, bad = model.tile(P)
        assert not bad and used > 0
        P += used
        if eob:
            break
    return bytes(model.out)


def test_tile_algorithm_matches_zlib_on_ultrafast_streams(golden_constants):
    r = np.random.default_rng(5)
    raws = [r.integers(0, 256, 20000, dtype=np.uint8).tobytes(), synth.gen_stream_np(0, 65536).tobytes(),
            synth.gen_stream_np(7, 65536).tobytes(), synth.gen_stream_np(3, 4096).tobytes(),
            bytes([5]) * 2048, bytes([128]) * 2048, b"Hello world! " * 9, bytes(70000),
            (r.integers(0, 256, 50000, dtype=np.uint8) % 5).astype(np.uint8).tobytes()]
    for raw in raws:
        comp = ob.compress_ultra_fast(raw)
        assert zlib.decompress(comp) == raw
        assert _decode_ultrafast_with_tiles(comp, golden_constants) == raw
