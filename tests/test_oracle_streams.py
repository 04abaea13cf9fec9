"""The generated test streams themselves, checked on the CPU against the oracle and zlib:
valid streams decode to their payload, crafted malformed streams produce the intended error."""
import zlib

import oracle_binding as ob
import streams


def test_valid_streams_decode_with_oracle_and_zlib():
    for name, comp, raw in streams.valid_streams():
        cap = max(len(raw), 1)
        if name == "fixed_sym286_is_eob":
            # with the slot exactly full the reference only recognises the real EOB code
            # (src/decompress.rs:1009-1015), so symbol 286 needs one spare byte of room
            assert ob.STATUS_NAMES[ob.decompress_bounded(comp, cap)[0]] == "OutputTooLarge"
            cap += 1
        st, out, ad = ob.decompress_bounded(comp, cap)
        assert st == 0, (name, ob.STATUS_NAMES[st])
        assert out == raw, name
        assert ad == zlib.adler32(raw)
        if name != "fixed_sym286_is_eob":  # zlib rejects 286/287 (parity trap 1)
            assert zlib.decompress(comp) == raw, name
        else:
            try:
                zlib.decompress(comp)
                assert False, "zlib should reject symbol 286"
            except zlib.error:
                pass


def test_error_streams_hit_the_intended_error():
    seen = set()
    for name, comp, expect in streams.error_streams():
        st, _, _ = ob.decompress_bounded(comp, 1 << 16)
        assert ob.STATUS_NAMES[st] == expect, (name, ob.STATUS_NAMES[st], expect)
        seen.add(expect)
    # every reachable error kind is covered
    assert seen >= {"BadZlibHeader", "InsufficientInput", "InvalidBlockType",
                    "InvalidUncompressedBlockLength", "InvalidHlit", "InvalidHdist",
                    "InvalidCodeLengthRepeat", "BadCodeLengthHuffmanTree",
                    "BadLiteralLengthHuffmanTree", "BadDistanceHuffmanTree", "InvalidDistanceCode",
                    "DistanceTooFarBack", "WrongChecksum"}


def test_mutations_never_crash_oracle_and_match_zlib_when_ok():
    n_ok = 0
    for name, comp in streams.mutation_streams():
        st, out, _ = ob.decompress_bounded(comp, 1 << 16)
        if st == 0:
            n_ok += 1
            assert zlib.decompress(comp) == out, name
    assert n_ok >= 0


def test_chunking_invariance_on_all_valid_streams():
    for name, comp, raw in streams.valid_streams():
        if len(comp) > 20000:
            continue
        for chunk in (1, 2, 5):
            if len(comp) // chunk > 4500:
                continue
            st, out = ob.decompress_by_chunks(comp, chunk)
            assert st == 0 and out == raw, (name, chunk)
