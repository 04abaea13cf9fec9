"""Test-stream factory shared by the CPU and GPU suites: valid streams of every block type,
hand-crafted malformed streams for every reachable DecompressionError, and mutation fuzz.

Nothing here reads /root/reference at run time (fixtures live in tests/golden/).
"""
import os
import zlib

import numpy as np

import oracle_binding as ob
from fdeflate_amd import synth

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


# --------------------------------------------------------------------------------------
# bit-level deflate writer for hand-made blocks
# --------------------------------------------------------------------------------------
class BitWriter:
    def __init__(self):
        self.acc = 0
        self.n = 0

    def bits(self, value, nbits):
        self.acc |= (value & ((1 << nbits) - 1)) << self.n
        self.n += nbits
        return self

    def align(self):
        self.n = (self.n + 7) & ~7
        return self

    def raw(self, data):
        self.align()
        for b in data:
            self.bits(b, 8)
        return self

    def tobytes(self):
        return self.acc.to_bytes((self.n + 7) // 8, "little")


def canonical_codes(lengths):
    """RFC 1951 3.2.2 -> {sym: (bit-reversed code, length)}"""
    codes = {}
    code = 0
    for ln in range(1, 16):
        for s, l in enumerate(lengths):
            if l == ln:
                rev = int(format(code, "0%db" % ln)[::-1], 2)
                codes[s] = (rev, ln)
                code += 1
        code <<= 1
    return codes


LEN_BASE = [3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115,
            131, 163, 195, 227, 258]
LEN_EXTRA = [0] * 8 + [1] * 4 + [2] * 4 + [3] * 4 + [4] * 4 + [5] * 4 + [0]
DIST_BASE = [1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537,
             2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577]
DIST_EXTRA = [0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13]
CLCL_ORDER = [16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15]

# complete code-length code: symbols 0..13 -> 4 bits, 14..17 -> 5 bits, 18 unused
CL_LENGTHS_DEFAULT = [4] * 14 + [5] * 4 + [0]
# variant that can also emit symbol 18: 0..12 -> 4 bits, 13..18 -> 5 bits (13/16 + 6/32 = 1)
CL_LENGTHS_WITH_18 = [4] * 13 + [5] * 6


def dynamic_header(w, litlen_lengths, dist_lengths, final=True, cl_lengths=None, cl_syms=None,
                   hlit=None, hdist=None):
    """Writes a dynamic block header.  cl_syms: optional explicit list of (symbol, extra_value)
    code-length symbols; default = one literal length symbol per code length."""
    cl_lengths = list(cl_lengths or CL_LENGTHS_WITH_18)
    hlit = len(litlen_lengths) if hlit is None else hlit
    hdist = len(dist_lengths) if hdist is None else hdist
    w.bits(1 if final else 0, 1).bits(2, 2)
    w.bits(hlit - 257, 5).bits(hdist - 1, 5)
    hclen = 19
    w.bits(hclen - 4, 4)
    for i in range(hclen):
        w.bits(cl_lengths[CLCL_ORDER[i]], 3)
    cc = canonical_codes(cl_lengths)
    if cl_syms is None:
        cl_syms = [(l, 0) for l in list(litlen_lengths) + list(dist_lengths)]
    for sym, extra in cl_syms:
        code, ln = cc[sym]
        w.bits(code, ln)
        if sym == 16:
            w.bits(extra, 2)
        elif sym == 17:
            w.bits(extra, 3)
        elif sym == 18:
            w.bits(extra, 7)
    return w


class DynBlock:
    """Helper to emit symbols of a dynamic block with given code lengths."""

    def __init__(self, w, litlen_lengths, dist_lengths, final=True, **kw):
        self.w = dynamic_header(w, litlen_lengths, dist_lengths, final, **kw)
        self.lc = canonical_codes(litlen_lengths)
        self.dc = canonical_codes(dist_lengths)
        if sum(1 for l in dist_lengths if l) == 1:  # single distance code: 1 bit, value 0
            s = [i for i, l in enumerate(dist_lengths) if l][0]
            self.dc = {s: (0, 1)}

    def lit(self, b):
        c, l = self.lc[b]
        self.w.bits(c, l)
        return self

    def eob(self):
        return self.lit(256)

    def match(self, length, dist, raw_dist_sym=None, raw_dist_bit=None):
        ls = max(i for i in range(29) if LEN_BASE[i] <= length and (i != 28 or length == 258))
        if length == 258:
            ls = 28
        c, l = self.lc[257 + ls]
        self.w.bits(c, l).bits(length - LEN_BASE[ls], LEN_EXTRA[ls])
        if raw_dist_bit is not None:
            self.w.bits(raw_dist_bit[0], raw_dist_bit[1])
            return self
        ds = raw_dist_sym if raw_dist_sym is not None else max(i for i in range(30) if DIST_BASE[i] <= dist)
        c, l = self.dc[ds]
        self.w.bits(c, l)
        if ds < 30:
            self.w.bits(dist - DIST_BASE[ds], DIST_EXTRA[ds])
        return self


def zlib_wrap(deflate_bytes, payload, header=b"\x78\x01", adler=None):
    a = zlib.adler32(payload) if adler is None else adler
    return header + deflate_bytes + a.to_bytes(4, "big")


def flat_lengths(nlit=286):
    """A complete litlen code: 256 literals + EOB + length symbols, all <= 9 bits."""
    # 286 symbols: 226 of 8 bits (226/256) + 60 of 9 bits (60/512 = 30/256) = 1
    return [8] * 226 + [9] * 60 if nlit == 286 else None


def long_code_lengths():
    """A complete litlen code with 13/14/15-bit codes (secondary tables in the reference):
    a handful of short codes plus a long tail."""
    lens = [0] * 286
    # lopsided prefix: 1,2,...  for some symbols, then fill
    # symbols 0..9: lengths 2,3,4,5,6,7,8,9,10,11 ; then the remaining space 2^-11 is split
    for i, l in enumerate([2, 3, 4, 5, 6, 7, 8, 9, 10, 11]):
        lens[i] = l
    # remaining codespace: 1 - sum_{l=2..11} 2^-l = 2^-1 + 2^-11 ; use symbol 10 with 1 bit
    lens[10] = 1
    # now remaining 2^-11 -> split over 16 symbols of length 15 (16 * 2^-15 = 2^-11)
    tail = [256, 257, 258, 259, 260, 264, 265, 270, 11, 12, 13, 14, 15, 16, 17, 285]
    for s in tail:
        lens[s] = 15
    assert abs(sum(2.0 ** -l for l in lens if l) - 1.0) < 1e-12
    return lens


def mixed_long_lengths():
    """Complete 286-symbol code mixing 7-, 8-, 12-, 13- and 14-bit codes, with some literals on
    the 13/14-bit codes: 240 x 8 bits (15/16) + 7 x 7 bits (7/128) + 27 x 12 + 8 x 13 + 4 x 14
    bits (together 1/128)."""
    lens = [8] * 240 + [7] * 7 + [12] * 27 + [13] * 8 + [14] * 4
    assert len(lens) == 286
    for a, b in ((5, 284), (6, 285), (7, 278), (8, 279), (9, 280), (10, 281)):
        lens[a], lens[b] = lens[b], lens[a]
    assert abs(sum(2.0 ** -l for l in lens if l) - 1.0) < 1e-12
    return lens


# --------------------------------------------------------------------------------------
# valid streams
# --------------------------------------------------------------------------------------
def rng(seed):
    return np.random.default_rng(seed)


def valid_streams(scale=1):
    """-> list of (name, compressed bytes, raw bytes)"""
    r = rng(1234)
    out = []

    def add(name, comp, raw):
        out.append((name, bytes(comp), bytes(raw)))

    datas = {
        "empty": b"",
        "hello": b"Hello world!",
        "one": b"\x07",
        "zeros2048": bytes(2048),
        "fives2048": bytes([5]) * 2048,
        "b128": bytes([128]) * 2048,
        "b254": bytes([254]) * 2048,
        "mod5": (r.integers(0, 256, 50000 * scale, dtype=np.uint8) % 5).astype(np.uint8).tobytes(),
        "uniform": r.integers(0, 256, 20000, dtype=np.uint8).tobytes(),
        "modelD_4k": synth.gen_stream_np(3, 4096).tobytes(),
        "modelD_64k": synth.gen_stream_np(0, 65536).tobytes(),
        "modelD_halfzero": synth.gen_stream_np(7, 65536).tobytes(),
        "allzero_64k": synth.gen_stream_np(15, 65536).tobytes(),
        "modelM": synth.gen_stream_np(1, 30000, "M").tobytes(),
        "modelL": synth.gen_stream_np(2, 30011, "L").tobytes(),
        "text": (b"the quick brown fox jumps over the lazy dog. " * 400),
    }
    blob = r.integers(0, 256, 33000, dtype=np.uint8).tobytes()
    datas["far_match"] = blob[:32760] + blob[:32760] + blob[5:2000]
    sparse = r.integers(0, 256, 70000, dtype=np.uint8)
    sparse[r.random(70000) < 0.9] = 0
    datas["sparse"] = sparse.tobytes()

    for name, d in datas.items():
        add("uf_" + name, ob.compress_ultra_fast(d), d)
        for level in (1, 6, 9):
            add("z%d_%s" % (level, name), zlib.compress(d, level), d)
        add("stored_" + name, ob.compress_stored(d), d)
        add("z0_" + name, zlib.compress(d, 0), d)
        for sname, strat in (("fixed", zlib.Z_FIXED), ("rle", zlib.Z_RLE), ("huff", zlib.Z_HUFFMAN_ONLY)):
            c = zlib.compressobj(6, zlib.DEFLATED, 15, 9, strat)
            add("%s_%s" % (sname, name), c.compress(d) + c.flush(), d)
    # small-window / odd zlib headers
    for wbits in (9, 12):
        c = zlib.compressobj(6, zlib.DEFLATED, wbits)
        d = datas["text"]
        add("wbits%d_text" % wbits, c.compress(d) + c.flush(), d)
    # sync / full / partial flushes: empty stored + empty fixed blocks, multi-block
    c = zlib.compressobj(6)
    comp = c.compress(b"abc") + c.flush(zlib.Z_SYNC_FLUSH) + c.flush(zlib.Z_FULL_FLUSH)
    comp += c.compress(datas["text"]) + c.flush(zlib.Z_SYNC_FLUSH) + c.compress(datas["mod5"][:3000]) + c.flush()
    add("multiflush", comp, b"abc" + datas["text"] + datas["mod5"][:3000])
    c = zlib.compressobj(6, zlib.DEFLATED, 15, 9, zlib.Z_FIXED)
    comp = c.compress(b"") + c.flush(zlib.Z_PARTIAL_FLUSH) + c.flush(zlib.Z_PARTIAL_FLUSH) + c.compress(b"xyz") + c.flush()
    add("partialflush_fixed", comp, b"xyz")
    # level-1 empty KAT with ten empty stored blocks spliced in (src/decompress.rs:1309-1325)
    spl = bytearray(ob.compress_stored(b""))
    for _ in range(10):
        spl[2:2] = bytes([0, 0, 0, 0xFF, 0xFF])
    add("zero_length_spliced", spl, b"")
    # hand-made dynamic blocks: long (13-15 bit) codes -> the reference's secondary tables
    for nm, lens in (("long15", long_code_lengths()), ("mixed_long", mixed_long_lengths())):
        usable = [s for s in range(256) if lens[s]]
        seq = r.choice(usable, size=3000)
        w = BitWriter()
        blk = DynBlock(w, lens, [1, 1])
        out_bytes = bytearray()
        for i, b in enumerate(seq):
            blk.lit(int(b))
            out_bytes.append(int(b))
            if i % 500 == 499 and lens[257]:
                blk.match(3, 2)
                out_bytes += bytes([out_bytes[-2], out_bytes[-1], out_bytes[-2]])
        blk.eob()
        add("dyn_" + nm, zlib_wrap(w.tobytes(), bytes(out_bytes)), out_bytes)
    # long distance codes (up to 15 bits) with real far matches
    w = BitWriter()
    dl = [1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 15] + [0] * 14
    lens = flat_lengths()
    blk = DynBlock(w, lens, dl)
    ob_ = bytearray()
    for i in range(2000):
        b = int(r.integers(0, 256))
        blk.lit(b)
        ob_.append(b)
        if i > 200 and i % 37 == 0:
            ds = int(r.integers(0, 16))
            dist = DIST_BASE[ds] + int(r.integers(0, 1 << DIST_EXTRA[ds]))
            dist = min(dist, len(ob_))
            ds2 = max(k for k in range(30) if DIST_BASE[k] <= dist)
            if ds2 < 16:
                ln = int(r.integers(3, 40))
                blk.match(ln, dist)
                for _ in range(ln):
                    ob_.append(ob_[-dist])
    blk.eob()
    add("dyn_longdist", zlib_wrap(w.tobytes(), bytes(ob_)), ob_)
    # fixed block using symbols 286/287 as end-of-block (parity trap 1): not valid for zlib
    w = BitWriter()
    w.bits(1, 1).bits(1, 2)
    fc = canonical_codes([8] * 144 + [9] * 112 + [7] * 24 + [8] * 8)
    for ch in b"trap":
        w.bits(*fc[ch])
    w.bits(*fc[286])
    add("fixed_sym286_is_eob", zlib_wrap(w.tobytes(), b"trap"), b"trap")
    return out


def corpus_streams():
    out = []
    d = os.path.join(GOLDEN, "vectors", "corpus")
    for name in sorted(os.listdir(d)):
        out.append(("corpus_" + name[:8], open(os.path.join(d, name), "rb").read()))
    for name in sorted(os.listdir(os.path.join(GOLDEN, "vectors"))):
        if name.endswith(".zz"):
            out.append(("zz_" + name[27:-3], open(os.path.join(GOLDEN, "vectors", name), "rb").read()))
    return out


# --------------------------------------------------------------------------------------
# malformed streams: one (or more) per reachable error
# --------------------------------------------------------------------------------------
def error_streams():
    """-> list of (name, bytes, expected status name or None when only parity matters)"""
    out = []
    good = zlib.compress(b"Hello world! " * 20, 6)
    out.append(("hdr_cm", b"\x77\x01" + good[2:], "BadZlibHeader"))
    out.append(("hdr_cinfo", b"\x88\x1c" + good[2:], "BadZlibHeader"))
    out.append(("hdr_fdict", b"\x78\x20" + good[2:], "BadZlibHeader"))
    out.append(("hdr_fcheck", b"\x78\x02" + good[2:], "BadZlibHeader"))
    out.append(("block_type3", b"\x78\x01" + BitWriter().bits(1, 1).bits(3, 2).bits(0, 13).tobytes() + bytes(8), "InvalidBlockType"))
    out.append(("stored_nlen", b"\x78\x01\x01\x05\x00\xfa\xfe" + b"abcde" + bytes(4), "InvalidUncompressedBlockLength"))
    w = BitWriter().bits(1, 1).bits(2, 2).bits(30, 5).bits(0, 5).bits(0, 4).bits(0, 16)
    out.append(("hlit_287", b"\x78\x01" + w.tobytes() + bytes(8), "InvalidHlit"))
    w = BitWriter().bits(1, 1).bits(2, 2).bits(0, 5).bits(30, 5).bits(0, 4).bits(0, 16)
    out.append(("hdist_31", b"\x78\x01" + w.tobytes() + bytes(8), "InvalidHdist"))
    lens = flat_lengths()
    # repeat-previous as first code length
    w = BitWriter()
    dynamic_header(w, lens, [1, 1], cl_syms=[(16, 0)] + [(8, 0)] * 10)
    out.append(("repeat_first", b"\x78\x01" + w.tobytes() + bytes(8), "InvalidCodeLengthRepeat"))
    # repeat running past hlit + hdist
    w = BitWriter()
    dynamic_header(w, lens, [1, 1], cl_syms=[(8, 0)] * 280 + [(18, 127)])
    out.append(("repeat_overrun", b"\x78\x01" + w.tobytes() + bytes(8), "InvalidCodeLengthRepeat"))
    # incomplete code-length code
    w = BitWriter()
    dynamic_header(w, lens, [1, 1], cl_lengths=[4] * 13 + [5] * 5 + [0])
    out.append(("bad_cl_tree", b"\x78\x01" + w.tobytes() + bytes(8), "BadCodeLengthHuffmanTree"))
    # incomplete litlen code -> BadCodeLengthHuffmanTree (sic, parity trap 2)
    bad = list(lens)
    bad[0] = 9
    w = BitWriter()
    dynamic_header(w, bad, [1, 1])
    out.append(("litlen_incomplete", b"\x78\x01" + w.tobytes() + bytes(8), "BadCodeLengthHuffmanTree"))
    over = list(lens)
    over[285] = 8
    w = BitWriter()
    dynamic_header(w, over, [1, 1])
    out.append(("litlen_oversubscribed", b"\x78\x01" + w.tobytes() + bytes(8), "BadCodeLengthHuffmanTree"))
    # no end-of-block code -> BadLiteralLengthHuffmanTree
    noeob = [0] * 286  # 256 literals x 8 bits is complete; symbol 256 has no code
    for s in range(256):
        noeob[s] = 8
    w = BitWriter()
    dynamic_header(w, noeob, [1, 1])
    out.append(("no_eob", b"\x78\x01" + w.tobytes() + bytes(8), "BadLiteralLengthHuffmanTree"))
    # bad distance tree: incomplete (3 codes of 2 bits), and a lone 2-bit code
    for nm, dl in (("dist_incomplete", [2, 2, 2]), ("dist_lone2", [2]), ("dist_over", [1, 1, 1])):
        w = BitWriter()
        dynamic_header(w, lens, dl)
        out.append((nm, b"\x78\x01" + w.tobytes() + bytes(8), "BadDistanceHuffmanTree"))
    # invalid distance code: all-zero distance lengths but a match is used
    w = BitWriter()
    blk = DynBlock(w, lens, [0])
    blk.lit(65).lit(66).lit(67)
    c, l = blk.lc[257]
    w.bits(c, l).bits(0, 12)
    blk.eob()
    out.append(("dist_none_used", zlib_wrap(w.tobytes(), b"ABC") + bytes(4), "InvalidDistanceCode"))
    # single 1-bit distance code, stream uses the other bit value
    w = BitWriter()
    blk = DynBlock(w, lens, [1])
    blk.lit(65).lit(66).lit(67).match(3, 1, raw_dist_bit=(1, 1))
    blk.eob()
    out.append(("dist_single_other_bit", zlib_wrap(w.tobytes(), b"ABC") + bytes(4), "InvalidDistanceCode"))
    # distance symbols 30/31 in the fixed code
    for sym in (30, 31):
        w = BitWriter().bits(1, 1).bits(1, 2)
        fc = canonical_codes([8] * 144 + [9] * 112 + [7] * 24 + [8] * 8)
        for ch in b"ABCD":
            w.bits(*fc[ch])
        w.bits(*fc[257])
        w.bits(int(format(sym, "05b")[::-1], 2), 5)
        w.bits(*fc[256])
        out.append(("fixed_dist%d" % sym, zlib_wrap(w.tobytes(), b"ABCD") + bytes(4), "InvalidDistanceCode"))
    # distance too far back
    w = BitWriter()
    blk = DynBlock(w, lens, [3] * 8)
    blk.lit(65).lit(66).match(3, 2).match(5, 8)
    blk.eob()
    out.append(("too_far_back", zlib_wrap(w.tobytes(), b"ABABA") + bytes(4), "DistanceTooFarBack"))
    w = BitWriter()
    blk = DynBlock(w, lens, [1, 1])
    blk.match(3, 1)
    blk.eob()
    out.append(("starts_with_run", zlib_wrap(w.tobytes(), b"") + bytes(4), "DistanceTooFarBack"))
    # wrong checksum
    bad = bytearray(good)
    bad[-1] ^= 0x55
    out.append(("wrong_checksum", bytes(bad), "WrongChecksum"))
    bad = bytearray(ob.compress_ultra_fast(synth.gen_stream_np(5, 4096).tobytes()))
    bad[-3] ^= 0x01
    out.append(("wrong_checksum_uf", bytes(bad), "WrongChecksum"))
    # truncations
    for nm, comp in (("z6", good), ("uf", ob.compress_ultra_fast(b"Hello world! " * 20)),
                     ("stored", ob.compress_stored(b"Hello world! " * 20))):
        for cut in sorted(set([0, 1, 2, 3, 5, len(comp) // 2, len(comp) - 5, len(comp) - 4, len(comp) - 1])):
            out.append(("trunc_%s_%d" % (nm, cut), comp[:cut], "InsufficientInput"))
    return out


def mutation_streams(n_per_seed=40, seeds=(1, 2, 3)):
    """Random corruptions of valid streams: only parity with the oracle matters."""
    out = []
    bases = []
    bases.append(zlib.compress((b"abcdefgh" * 300) + bytes(range(256)), 6))
    bases.append(zlib.compress(synth.gen_stream_np(9, 3000).tobytes(), 9))
    bases.append(ob.compress_ultra_fast(synth.gen_stream_np(4, 2000).tobytes()))
    c = zlib.compressobj(6, zlib.DEFLATED, 15, 9, zlib.Z_FIXED)
    bases.append(c.compress(b"fixed huffman " * 50) + c.flush())
    bases.append(zlib.compress(bytes(700), 0))
    for seed in seeds:
        r = rng(seed)
        for bi, base in enumerate(bases):
            for k in range(n_per_seed):
                b = bytearray(base)
                nflip = int(r.integers(1, 4))
                for _ in range(nflip):
                    pos = int(r.integers(0, len(b)))
                    if r.random() < 0.5:
                        b[pos] ^= 1 << int(r.integers(0, 8))
                    else:
                        b[pos] = int(r.integers(0, 256))
                if r.random() < 0.3:
                    b = b[:int(r.integers(1, len(b)))]
                out.append(("mut_s%d_b%d_%d" % (seed, bi, k), bytes(b)))
    return out


def pack_exact(blobs):
    """Back-to-back packing (arbitrary alignment): off[i+1] - off[i] == len(blob i)."""
    off = np.zeros(len(blobs) + 1, dtype=np.uint64)
    for i, b in enumerate(blobs):
        off[i + 1] = off[i] + len(b)
    buf = np.zeros(max(int(off[-1]), 1), dtype=np.uint8)
    for i, b in enumerate(blobs):
        buf[int(off[i]):int(off[i + 1])] = np.frombuffer(b, dtype=np.uint8)
    return buf, off
