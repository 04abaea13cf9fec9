"""-m gpu: the streaming `Decompressor` (fdh_decompressor_*, reference src/decompress.rs:158-337)
through ctypes against the oracle's restatement of the same object.

The harness below is the reference's own `decompress_by_chunks`
(src/decompress/tests/test_utils.rs:47-87) written against the product; the oracle side is
`fdo_decompress_by_chunks`.  The remaining tests restate the reference's unit tests
(src/decompress.rs:1261-1384) and the logic of its resumability fuzz targets
(fuzz/fuzz_targets/inflate_bytewise{,2,3}.rs, inflate_split.rs)."""
import itertools
import zlib

import numpy as np
import pytest

import oracle_binding as ob
import streams

pytestmark = pytest.mark.gpu

TOO_MANY_ITERATIONS = -2
TEST_OUTPUT_TOO_LARGE = -1


@pytest.fixture(scope="module")
def fd():
    import torch
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    import fdeflate_amd
    return fdeflate_amd


def product_by_chunks(fd, data, chunks, out_cap=1_000_000):
    """test_utils.rs:47-87 on fdeflate_amd.Decompressor -> (status, bytes)."""
    chunks = iter(chunks)
    d = fd.Decompressor()
    d.ignore_adler32()
    out = bytearray(out_cap)
    in_pos = out_pos = 0
    it = 0
    while not d.is_done():
        it += 1
        if it > 5000:
            return TOO_MANY_ITERATIONS, b""
        size = next(chunks, 0)
        end = min(in_pos + size, len(data))
        try:
            consumed, produced = d.read(data[in_pos:end], out, out_pos)
        except fd.DecompressionError as e:
            return e.status, b""
        in_pos += consumed
        out_pos += produced
        if out_pos == len(out) and consumed == 0 and not d.is_done():
            return TEST_OUTPUT_TOO_LARGE, b""
    return 0, bytes(out[:out_pos])


def oracle_by_chunks(data, chunk):
    st, out = ob.decompress_by_chunks(data, chunk)
    return st, (out if st == 0 else b"")


def test_zz_vectors_whole_and_bytewise(fd, golden_manifest):
    """src/decompress.rs:1331-1384: example1 -> 281 bytes, Adler-32 751299; example2/3 ->
    BadLiteralLengthHuffmanTree, each whole-input and byte-at-a-time."""
    items = dict(streams.corpus_streams())
    d1 = items["zz_example1"]
    for chunks in ([len(d1)], itertools.repeat(1)):
        st, out = product_by_chunks(fd, d1, chunks)
        assert st == 0 and len(out) == 281 and zlib.adler32(out) == 751299
    for name in ("zz_example2", "zz_example3"):
        d = items[name]
        for chunks in ([len(d)], itertools.repeat(1)):
            st, _ = product_by_chunks(fd, d, chunks)
            assert ob.STATUS_NAMES[st] == "BadLiteralLengthHuffmanTree", (name, st)


def test_corpus_chunking_invariance_vs_oracle(fd, golden_manifest):
    """Every corpus stream and .zz vector: whole input, 3-byte and 7-byte chunks -- the same result
    as the oracle's decompress_by_chunks with the same chunking; byte-at-a-time on a sample."""
    items = streams.corpus_streams()
    for k, (name, data) in enumerate(items):
        plans = [0, 7, 3] + ([1] if (k % 11 == 0 or name.startswith("zz_")) else [])
        for chunk in plans:
            exp = oracle_by_chunks(data, chunk)
            chunks = [len(data)] if chunk == 0 else itertools.repeat(chunk)
            got = product_by_chunks(fd, data, chunks)
            assert got == exp, (name, chunk, got[0], exp[0])
        if name.startswith("corpus_"):
            full = [n for n in golden_manifest["corpus"] if n.startswith(name[7:])][0]
            exp = golden_manifest["corpus"][full]
            st, out = product_by_chunks(fd, data, [len(data)])
            assert (st, len(out), zlib.adler32(out)) == (0, exp["length"], exp["adler32"])


def test_every_block_type_in_ragged_chunks(fd):
    """stored / fixed / dynamic / multi-block / 15-bit-code streams fed in ragged chunks."""
    r = np.random.default_rng(5)
    for name, comp, raw in streams.valid_streams():
        if len(comp) > 6000:
            continue
        sizes = [int(x) for x in r.integers(0, 97, size=400)]
        st, out = product_by_chunks(fd, comp, itertools.chain(sizes, itertools.repeat(64)))
        assert (st, out) == (0, raw), name


def test_error_and_mutated_streams_chunked_vs_oracle(fd):
    items = [(n, b) for n, b, _ in streams.error_streams()]
    items += streams.mutation_streams(n_per_seed=12, seeds=(21,))
    for name, data in items:
        for chunk in (0, 5):
            exp = oracle_by_chunks(data, chunk)
            got = product_by_chunks(fd, data, [len(data)] if chunk == 0 else itertools.repeat(chunk))
            # errors of one input are the same for every chunking (inflate_split.rs / inflate_bytewise3.rs)
            assert got == exp, (name, chunk, got[0], exp[0])


def test_checksum_after_eof(fd):
    """src/decompress.rs:1282-1307: the last checksum byte arrives in a later call -> (len-1, 12)
    then (1, 0) and done."""
    payload = b"Hello world!"
    comp = zlib.compress(payload, 1)
    d = fd.Decompressor()
    buf = bytearray(1024)
    c, p = d.read(comp[:-1], buf, 0)
    assert (c, p) == (len(comp) - 1, len(payload)) and not d.is_done()
    out = bytearray(buf[:p])
    c2, p2 = d.read(comp[c:], out, p)
    assert d.is_done() and (c2, p2) == (1, 0)
    assert bytes(out) == payload
    assert d.read(b"more", out, p) == (0, 0)       # :185-187


def test_wrong_checksum_and_ignore_adler32(fd):
    """src/decompress.rs:1261-1280."""
    comp = bytearray(zlib.compress(b"Hello world!", 1))
    comp[-1] = (comp[-1] + 1) & 0xFF
    with pytest.raises(fd.DecompressionError) as ei:
        fd.decompress_to_vec(bytes(comp))
    assert ei.value.kind == "WrongChecksum"
    d = fd.Decompressor()
    with pytest.raises(fd.DecompressionError) as ei:
        d.read(bytes(comp), bytearray(1024), 0)
    assert ei.value.kind == "WrongChecksum"
    d = fd.Decompressor()
    d.ignore_adler32()
    buf = bytearray(1024)
    c, p = d.read(bytes(comp), buf, 0)
    assert bytes(buf[:p]) == b"Hello world!" and d.is_done()


def test_zero_length(fd):
    """src/decompress.rs:1309-1325: ten empty stored blocks spliced into compress_to_vec(b""),
    zero-length output slice."""
    comp = bytearray(bytes.fromhex("7801030000000001"))   # level-1 compress_to_vec(b"") (compress/mod.rs:71,234-238)
    for _ in range(10):
        comp[2:2] = bytes([0, 0, 0, 0xFF, 0xFF])
    d = fd.Decompressor()
    c, p = d.read(bytes(comp), bytearray(0), 0)
    assert d.is_done() and (c, p) == (len(comp), 0)


def test_one_byte_output_windows(fd):
    """fuzz_targets/inflate_bytewise.rs: the output grows one byte per call and every call
    produces exactly one byte (:35)."""
    r = np.random.default_rng(3)
    for level in (0, 1, 6, 9):
        data = bytes(r.integers(0, 7, size=150, dtype=np.uint8)) + b"abcabcabcabc" * 5
        comp = zlib.compress(data, level)
        d = fd.Decompressor()
        out = bytearray()
        in_pos = 0
        while True:
            pos = len(out)
            if pos < len(data):
                out.append(1)
            consumed, produced = d.read(comp[in_pos:], out, pos)
            in_pos += consumed
            assert produced == 1, (level, pos)
            if d.is_done():
                break
        assert bytes(out) == data


def test_one_byte_input_feeds(fd):
    """fuzz_targets/inflate_bytewise2.rs: one input byte per call, 1 KiB of room."""
    r = np.random.default_rng(4)
    data = bytes(r.integers(0, 5, size=700, dtype=np.uint8))
    for level in (1, 6):
        comp = zlib.compress(data, level)
        d = fd.Decompressor()
        out = bytearray()
        pos = in_pos = 0
        while not d.is_done():
            out.extend(bytes(pos + 1024 - len(out)))
            consumed, produced = d.read(comp[in_pos:in_pos + 1], out, pos)
            in_pos += consumed
            pos += produced
            assert in_pos <= len(comp)
        assert bytes(out[:pos]) == data


def test_split_in_two_equals_one_shot(fd):
    """fuzz_targets/inflate_split.rs: decoding in two steps == decompress_to_vec, for valid,
    truncated and corrupted inputs."""
    r = np.random.default_rng(8)
    raw = bytes(r.integers(0, 9, size=5000, dtype=np.uint8))
    bases = [zlib.compress(raw, 6), ob.compress_ultra_fast(raw), zlib.compress(raw, 0)]
    cases = []
    for b in bases:
        cases.append(b)
        cases.append(b[:len(b) // 2])                 # truncated
        m = bytearray(b)
        m[len(m) // 3] ^= 0x5A
        cases.append(bytes(m))                        # corrupted
    for data in cases:
        try:
            full = (0, fd.decompress_to_vec(data))
        except fd.DecompressionError as e:
            full = (e.status, b"")
        for cut in (1, len(data) // 3, len(data) - 1):
            a, b = data[:cut], data[cut:]

            def run():
                d = fd.Decompressor()
                out = bytearray(1024)
                ii = oi = 0
                while not d.is_done() and ii < len(a):
                    c, p = d.read(a[ii:], out, oi)
                    ii += c
                    oi += p
                    if oi == len(out):
                        out.extend(bytes(32 * 1024))
                    assert c > 0 or p > 0 or d.is_done()
                while not d.is_done():
                    c, p = d.read(b[ii - len(a):], out, oi)
                    if not d.is_done() and c == 0 and p == 0:
                        return 2, b""                 # InsufficientInput
                    ii += c
                    oi += p
                    out.extend(bytes(oi + 32 * 1024 - len(out)))
                return 0, bytes(out[:oi])

            try:
                got = run()
            except fd.DecompressionError as e:
                got = (e.status, b"")
            assert got == full, (len(data), cut, got[0], full[0])


def test_partial_length_on_truncated_input(fd):
    """fdh_inflate_batch reports, for InsufficientInput, the bytes `read` had produced when the
    input ran out (the streaming object relies on it): every prefix of three streams against the
    oracle's Decompressor fed the same prefix in one call."""
    import gpu_harness
    bases = [zlib.compress(bytes(range(200)) * 3, 6), ob.compress_ultra_fast(b"Hello world! " * 9),
             ob.compress_stored(b"stored block payload " * 4)]
    blobs, exp = [], []
    for base in bases:
        for cut in range(len(base)):
            d = ob.Decompressor()
            out = np.zeros(4096, dtype=np.uint8)
            st, c, p = d.read(base[:cut], out, 0)
            assert st == 0 and not d.is_done()
            blobs.append(base[:cut])
            exp.append((p, out[:p].tobytes()))
    st, ln, ad, outs, ok = gpu_harness.gpu_inflate(blobs, [4096] * len(blobs))
    assert ok
    for i, (p, data) in enumerate(exp):
        assert int(st[i]) == 2, i
        assert int(ln[i]) == p and outs[i][:p].tobytes() == data, (i, int(ln[i]), p)


def test_c1_single_4k_stream_through_decompressor(fd):
    """BASELINE config 1: one 4 KiB model-D buffer in the ultra-fast format through
    Decompressor::read -- in one call, in 64-byte chunks and byte-at-a-time -- against the oracle's
    streaming decoder with the same chunking."""
    from fdeflate_amd import synth
    raw = synth.gen_stream_np(3, 4096, png_rows=False).tobytes()
    comp = ob.compress_ultra_fast(raw)
    for chunk in (0, 64, 1):
        exp = oracle_by_chunks(comp, chunk)
        got = product_by_chunks(fd, comp, [len(comp)] if chunk == 0 else itertools.repeat(chunk))
        assert exp == (0, raw) and got == exp, chunk


def test_bounded_output_with_history_compaction(fd):
    """The png-crate pattern (src/decompress.rs:158-178): a small output buffer, the consumer
    takes the bytes and keeps going from a new position."""
    r = np.random.default_rng(12)
    raw = bytes(r.integers(0, 4, size=200_000, dtype=np.uint8))
    comp = zlib.compress(raw, 6)
    d = fd.Decompressor()
    got = bytearray()
    buf = bytearray(40_000)
    pos = in_pos = 0
    guard = 0
    while not d.is_done():
        guard += 1
        assert guard < 400
        c, p = d.read(comp[in_pos:in_pos + 3000], buf, pos)
        in_pos += c
        got += buf[pos:pos + p]
        pos += p
        if pos > 32_768 + 4096:          # keep 32 KiB of history in front, as png does
            keep = 32_768
            buf[:keep] = buf[pos - keep:pos]
            pos = keep
    assert bytes(got) == raw


def test_attempts_go_on_where_the_last_one_stopped(fd):
    """The device-side resume points (fdh_inflate_batch_resumable): however the input arrives and however small
    the window, the attempts of one stream together decode little more than the stream once (rounds 1-3: every
    attempt started at the first byte -- 2 x with decode-ahead, O(N^2 / chunk) for input in small pieces).  Input
    in 20 KB pieces into a large buffer; everything at once through a 16 KiB window; both at once; for a zlib
    level-6 stream of many blocks and for a 1.5 MB stream in the ultra-fast format (one block, pairs of literals)."""
    r = np.random.default_rng(5)
    raw = bytes((np.cumsum(r.integers(-3, 4, size=1_500_000)) & 0xFF).astype(np.uint8))
    for comp in (zlib.compress(raw, 6), ob.compress_ultra_fast(raw)):
        # input in pieces, room for everything
        d = fd.Decompressor()
        buf = bytearray(len(raw) + 64)
        pos = 0
        for k in range(0, len(comp), 20_000):
            c, p = d.read(comp[k:k + 20_000], buf, pos)
            pos += p
        while not d.is_done():
            c, p = d.read(b"", buf, pos)
            pos += p
            assert p > 0 or d.is_done()
        assert bytes(buf[:pos]) == raw
        assert d.decoded_bytes() <= len(raw) * 1.05 + 70_000, (d.decoded_bytes(), len(raw), d.attempts())
        # everything at once through a small window, and both in pieces
        for piece in (len(comp), 50_000):
            d = fd.Decompressor()
            got = bytearray()
            buf = bytearray(32_768 + 16_384)
            pos = 0
            k = 0
            calls = 0
            while not d.is_done():
                calls += 1
                assert calls < 20_000
                chunk = comp[k:k + piece]
                c, p = d.read(chunk, buf, pos)
                k += c
                got += buf[pos:pos + p]
                pos += p
                if pos > 32_768:
                    buf[:32_768] = buf[pos - 32_768:pos]
                    pos = 32_768
            assert bytes(got) == raw
            assert d.decoded_bytes() <= len(raw) * 1.05 + 70_000, (piece, d.decoded_bytes(), len(raw), d.attempts())


def test_large_stream_through_a_small_window_decodes_ahead(fd):
    """A 3 MB multi-block stream with the whole input at hand, drained through a 16 KiB window (the
    png-crate pattern with history compaction): bytes and end state as the oracle's, and the number
    of decode attempts stays small -- every attempt decodes ahead of what the caller can take (up to 128 KiB: the
    bound that keeps the object's device memory under 1 MiB, test_device_memory_stays_at_the_reference_footprint)
    and the following calls are served from that prefix (one attempt per call would be ~190 decodes
    of ever longer prefixes).  Then the same with a wrong checksum: the error arrives with the last
    bytes, not before, and a hard error in the middle of the stream arrives only when the window
    gets there."""
    r = np.random.default_rng(77)
    raw = bytes((np.cumsum(r.integers(-3, 4, size=3_000_000)) & 0xFF).astype(np.uint8))
    comp = zlib.compress(raw, 6)

    def drain(data, window=16_384):
        d = fd.Decompressor()
        got = bytearray()
        buf = bytearray(32_768 + window)
        pos = 0
        k = 0
        status = 0
        calls = 0
        while not d.is_done():
            calls += 1
            assert calls < 5000
            try:
                c, p = d.read(data[k:], buf, pos)   # (what is not consumed is offered again: src/decompress.rs:167-170)
            except fd.DecompressionError as e:
                status = e.status
                break
            k += c
            got += buf[pos:pos + p]
            pos += p
            if p == 0 and c == 0 and pos < len(buf):
                break                      # nothing more without more input
            if pos > 32_768:               # keep 32 KiB of history in front, as png does
                buf[:32_768] = buf[pos - 32_768:pos]
                pos = 32_768
        return status, bytes(got), d.attempts(), calls

    st, got, attempts, calls = drain(comp)
    assert st == 0 and got == raw
    assert calls > 150 and attempts <= 32, (calls, attempts)
    # wrong checksum: every byte is delivered, the error comes with the end of the stream
    bad = bytearray(comp); bad[-1] ^= 0x55
    st, got, attempts, _ = drain(bytes(bad))
    # (like the reference's Err, the call that fails does not say how much it produced: its bytes are lost to
    # a caller that goes by the return value)
    assert fd.STATUS_NAMES[st] == "WrongChecksum" and attempts <= 32
    assert raw.startswith(got) and len(got) >= len(raw) - 16_384
    # a flipped bit in the middle: the bytes in front of the damage are delivered first
    mid = bytearray(comp); mid[len(comp) // 2] ^= 0x10
    st, got, attempts, _ = drain(bytes(mid))
    est, _, _ = ob.decompress_bounded(bytes(mid), 4_000_000)   # the one-shot classification of the whole input
    _, eout = ob.decompress_by_chunks(bytes(mid), 0, 4_000_000)   # ... and what the reference had produced by then
    assert st == est, (st, est)
    if est == 0:
        assert got == eout
    else:
        assert eout.startswith(got) and len(got) >= len(eout) - 16_384 and len(got) > 500_000
        # the damage is met once decoding ahead; from then on the attempts stay with the exact slot instead of
        # decoding twice per call (round-3 review): about one attempt per call behind that point, not two
        assert attempts <= 32 + 1 + (len(eout) // 2) // 16_384 + 8, attempts


def test_device_memory_stays_at_the_reference_footprint(fd):
    """The reference keeps its tables and the last 32 KiB of the caller's buffer (src/decompress.rs:96-113,
    1067-1070) however long the stream is.  So does the object: a 64 MB zlib stream of many blocks and a 32 MB
    stream in the ultra-fast format -- ONE block, whose header (parsed again by every attempt, for the tables) is
    tens of megabytes behind the decoder at the end -- drained through a 16 KiB window, input offered in 64 KiB
    pieces (what is not consumed is offered again): every byte right, device memory under 1 MiB throughout
    (fdh_decompressor_device_bytes: the high-water mark of the object's buffers), and the stream decoded once."""
    r = np.random.default_rng(11)
    for kind, n in (("zlib6", 64 << 20), ("ultrafast", 32 << 20)):
        raw = (np.cumsum(r.integers(-2, 3, size=n, dtype=np.int8), dtype=np.uint8)).tobytes()
        comp = zlib.compress(raw, 6) if kind == "zlib6" else ob.compress_ultra_fast(raw)
        d = fd.Decompressor()
        window = 16_384
        buf = bytearray(32_768 + window)
        pos = k = total = calls = 0
        view = memoryview(raw)
        while not d.is_done():
            calls += 1
            assert calls < 40_000, (kind, total, k)
            c, p = d.read(comp[k:k + 65_536], buf, pos)
            k += c
            assert buf[pos:pos + p] == view[total:total + p], (kind, total, p)
            total += p
            pos += p
            if pos > 32_768:                   # keep 32 KiB of history in front, as png does
                buf[:32_768] = buf[pos - 32_768:pos]
                pos = 32_768
            assert d.device_bytes() <= 1 << 20, (kind, d.device_bytes(), total)
        assert total == n and k == len(comp), (kind, total, k, len(comp))
        assert d.decoded_bytes() <= n * 1.02 + 500_000, (kind, d.decoded_bytes(), d.attempts())


def test_reference_bounded_loop_over_the_streaming_object(fd):
    """The loop of the reference's own `decompress_to_vec_bounded` (src/decompress.rs:1111-1144) written against
    `Decompressor.read` -- the whole remaining input offered every time, a growing output vector, "truncated" only
    once a read with input left has consumed and produced nothing while the output still has room -- on streams of
    more than 256 KiB of input, where the object does NOT consume everything on every call (include/fdeflate_hip.h:
    the unread input on the device is bounded) and does not attempt a decode on every call either.  The loop must
    terminate with every byte, for a zlib level-6 stream, an ultra-fast one and a stored one; and the same loop fed
    in 40 000-byte pieces (the caller keeps what was not consumed and offers it again with the next piece) -- the
    pattern round 5's review asked to see -- must too.  A cut stream ends InsufficientInput, as the reference's does."""
    r = np.random.default_rng(23)
    raw = (np.cumsum(r.integers(-3, 4, size=3_000_000, dtype=np.int8), dtype=np.uint8)).tobytes()
    streams_ = {"zlib6": zlib.compress(raw, 6), "ultrafast": ob.compress_ultra_fast(raw), "stored": zlib.compress(raw[:900_000], 0)}

    def bounded_loop(comp, maxlen, piece=None):
        d = fd.Decompressor()
        out = bytearray(min(1024, maxlen))
        out_pos = in_pos = offered = 0
        calls = 0
        while not d.is_done():
            calls += 1
            assert calls < 20_000
            offered = len(comp) if piece is None else min(len(comp), max(offered, in_pos) + piece)
            data = comp[in_pos:offered]
            try:
                c, p = d.read(data, out, out_pos)
            except fd.DecompressionError as e:
                return e.status, bytes(out[:out_pos])
            in_pos += c
            out_pos += p
            if c == 0 and p == 0 and out_pos < len(out) and offered == len(comp) and in_pos == len(comp) and not d.is_done():
                # everything has been handed over and nothing moves: one flush (an empty read) decides
                c2, p2 = d.read(b"", out, out_pos)
                out_pos += p2
                if p2 == 0 and not d.is_done():
                    return 2, bytes(out[:out_pos])  # InsufficientInput
            if out_pos == len(out) and not d.is_done():
                if len(out) >= maxlen:
                    return 17, bytes(out)
                out.extend(bytes(min(len(out), maxlen - len(out))))  # (doubling, src/decompress.rs:1139)
        return 0, bytes(out[:out_pos])

    for name, comp in streams_.items():
        want = zlib.decompress(comp)
        assert len(comp) > 262_144 or name == "zlib6", (name, len(comp))
        for piece in (None, 40_000):
            st, got = bounded_loop(comp, len(want), piece)
            assert st == 0 and got == want, (name, piece, st, len(got), len(want))
        cut = comp[:len(comp) * 3 // 4]
        st, got = bounded_loop(cut, len(want), 40_000)
        rst, rout = ob.decompress_bounded(cut, len(want))[:2]
        assert st == 2 == rst and want.startswith(got), (name, st, rst, len(got))


def test_no_resume_record_was_lost(fd):
    """Runs last in this file.  A kernel that meets a status promising a resume record over a record that reads all
    zero counts it (inflate.hip, g_lost_records: round 5 saw that with scratch from the device's default memory pool)
    and goes on from the caller's own record, never from the stream's first byte.  After everything above -- a few
    thousand resumed calls -- the count must still be zero: the scratch pool of the library's own does its job."""
    import ctypes as C
    from fdeflate_amd import _lib
    n = C.c_uint(123456)
    assert _lib.lib().fdh_debug_lost_records(C.byref(n)) == 0
    assert n.value == 0, n.value
