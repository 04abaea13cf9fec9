"""Executable CPU model of the counting pass of the landing decoder (fdeflate_amd/csrc/inflate_seg3.h).

One ultra-fast-format stream (reference src/compress/ultrafast.rs:82-181) is cut into 64 equal bit
segments.  Lane l > 0 walks a GUESSED chain of table look-ups (up to three literals per look-up) from
the first bit of its segment until it has left a window of kWindow bits: x0[l].  Every lane then
counts its bytes from x0[l] (lane 0: from the first token) and must LAND exactly on x0[l + 1]: whole
groups of look-ups while they cannot pass it, then single look-ups, and the first literal of a step
alone once the whole step would pass the target.  A run (a length symbol with the one distance code
of the prefix) is no table step: the lane that meets one follows the chain of run tokens and cuts its
interval behind it.  A lane that lands proves its right neighbour's guess, by induction from
lane 0.  This model states the rules; tests/test_seg3_model.py runs it against zlib.
"""
import numpy as np

K_BITS = 12
WINDOW = 256
GROUP = 8          # look-ups per group
METER = 32         # look-ups per interval


def canonical_codes(lengths):
    """RFC 1951 3.2.2: code of every symbol, MSB first."""
    max_len = max(lengths)
    bl_count = [0] * (max_len + 1)
    for l in lengths:
        if l:
            bl_count[l] += 1
    code = 0
    next_code = [0] * (max_len + 2)
    for bits in range(1, max_len + 1):
        code = (code + bl_count[bits - 1]) << 1
        next_code[bits] = code
    codes = []
    for l in lengths:
        if l:
            codes.append(next_code[l])
            next_code[l] += 1
        else:
            codes.append(0)
    return codes


def bit_reverse(v, n):
    r = 0
    for _ in range(n):
        r = (r << 1) | (v & 1)
        v >>= 1
    return r


def build_tables(lengths):
    """first[i] = (symbol, length) of the code that starts index i (12 bits, LSB first), or None;
    step[i] = (bits, number of literals): up to three literals that fit the index together, (0, 0) when
    the first symbol is no literal / does not fit."""
    codes = canonical_codes(lengths)
    first = [None] * (1 << K_BITS)
    for sym, l in enumerate(lengths):
        if l == 0 or l > K_BITS:
            continue
        rev = bit_reverse(codes[sym], l)
        for idx in range(rev, 1 << K_BITS, 1 << l):
            first[idx] = (sym, l)
    step = []
    for i in range(1 << K_BITS):
        used = n = 0
        for _ in range(3):
            f = first[i >> used]
            if f is None or f[0] >= 256 or used + f[1] > K_BITS:
                break
            used += f[1]
            n += 1
        step.append((used, n))
    return first, step


class Stream:
    def __init__(self, data, lengths):
        self.bits = np.unpackbits(np.frombuffer(bytes(data), dtype=np.uint8), bitorder="little")
        self.nbits = self.bits.size
        self.first, self.step = build_tables(lengths)
        self.pad = np.zeros(64, dtype=np.uint8)

    def index(self, pos):
        """The 12 stream bits from `pos` on as a table index (zeros behind the end)."""
        b = self.bits[pos:pos + K_BITS]
        if b.size < K_BITS:
            b = np.concatenate([b, self.pad[:K_BITS - b.size]])
        return int(b.dot(1 << np.arange(K_BITS)))


LEN_BASE = [3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258]
LEN_EXTRA = [0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0]
REPEAT = 8         # run tokens followed in one chain (the general writer's streams)
REPEAT_LEAN = 64   # ... of a stream the lean writer takes (round 6: its chains never enter the output image, however long)


def special(st, pos):
    """The token at `pos`: ('run', bits, length) | ('eob', bits, 0) | ('lit', bits, 0) | ('bad', 0, 0).
    A run is a length symbol, its extra bits and the one distance code the prefix declares ('0')."""
    f = st.first[st.index(pos)]
    if f is None:
        return ("bad", 0, 0)
    sym, l = f
    if sym < 256:
        return ("lit", l, 0)
    if sym == 256 or sym >= 286:
        return ("eob", l, 0)
    ex = LEN_EXTRA[sym - 257]
    extra = 0
    for k in range(ex):
        extra |= int(st.bits[pos + l + k]) << k if pos + l + k < st.nbits else 0
    dist_bit = int(st.bits[pos + l + ex]) if pos + l + ex < st.nbits else 0
    if dist_bit:
        return ("bad", 0, 0)
    return ("run", l + ex + 1, LEN_BASE[sym - 257] + extra)


def guess(st, start, leave):
    """Guessed chain from `start`: groups of look-ups until the chain is at or behind `leave`; a run
    token is stepped over, anything else that is no literal slides on by one bit.  Returns where the
    chain is then."""
    pos = start
    while pos < leave:
        for _ in range(GROUP):
            used, _n = st.step[st.index(pos)]
            if used == 0:
                kind, bits, _r = special(st, pos)
                pos += bits if kind == "run" else 1
                break
            pos += used
    return pos


def chain(st, pos, bound):
    """The run chain at `pos`: the first run token and, while they are 258 bytes long, up to st.repeat - 1
    more, none of them reaching past `bound` (None: no bound).  Returns (bytes, end position) or None."""
    total = 0
    for rep in range(getattr(st, "repeat", REPEAT)):
        kind, bits, run = special(st, pos)
        if kind != "run":
            if rep == 0:
                return None
            break
        if bound is not None and pos + bits > bound:
            if rep == 0:
                return None
            break
        total += run
        pos += bits
        if run != 258 or (bound is not None and pos >= bound):
            break
    return total, pos


def count_to(st, pos, target, stats):
    """Counts the bytes from `pos` to exactly `target` (None: to the end-of-block code).  Returns
    (bytes, end position, checkpoints, ok).  A checkpoint is (position, bytes so far); two
    consecutive ones are at most METER steps apart."""
    cnt = 0
    ck = [(pos, 0)]
    m = 0

    def meter(inc):
        nonlocal m
        if m + inc > METER:
            ck.append((pos, cnt))
            m = 0
        m += inc

    def run_here():
        """A lane that sits on a token that is no literal: a run chain ends its interval."""
        nonlocal pos, cnt, m
        r = chain(st, pos, target)
        if r is None:
            return False
        cnt += r[0]
        pos = r[1]
        ck.append((pos, cnt))
        m = 0
        stats["chains"] += 1
        return True

    # whole groups while they cannot pass the target
    while target is None or pos + GROUP * K_BITS <= target:
        meter(GROUP)
        parked = False
        for _ in range(GROUP):
            used, n = st.step[st.index(pos)]
            if used == 0:
                parked = True
                break
            pos += used
            cnt += n
        if parked:
            if special(st, pos)[0] == "eob":
                break
            if not run_here():
                return cnt, pos, ck, False
    if target is None:
        ok = special(st, pos)[0] == "eob"
        ck.append((pos, cnt))
        return cnt, pos, ck, ok
    # single look-ups; once a step could pass the target, its first literal alone
    while pos < target:
        d = target - pos
        used, n = st.step[st.index(pos)]
        if used == 0:
            if not run_here():
                return cnt, pos, ck, False
            continue
        meter(1)
        if used <= d:
            pos += used
            cnt += n
            stats["single"] += 1
            continue
        f = st.first[st.index(pos)]
        stats["first"] += 1
        if f[1] > d:
            return cnt, pos, ck, False  # no token ends at the target: the neighbour's guess was wrong
        pos += f[1]
        cnt += 1
    ck.append((pos, cnt))
    return cnt, pos, ck, True


def plan(data, lengths, canon_bits, nseg=64, repeat=REPEAT):
    """The counting pass of one stream.  Returns (total bytes, per-lane results, stats) or None when a
    lane did not land (the kernel then counts that lane's neighbour again / leaves the stream).
    `repeat`: the run tokens a chain may merge (REPEAT, or REPEAT_LEAN for a stream the lean writer takes)."""
    st = Stream(data, lengths)
    st.repeat = repeat
    data_bits = st.nbits - canon_bits
    seg = (data_bits + nseg - 1) // nseg
    x0 = [canon_bits] + [guess(st, canon_bits + l * seg, canon_bits + l * seg + WINDOW) for l in range(1, nseg)]
    stats = {"single": 0, "first": 0, "chains": 0, "fail": 0}
    total = 0
    lanes = []
    for l in range(nseg):
        target = x0[l + 1] if l + 1 < nseg else None
        cnt, end, ck, ok = count_to(st, x0[l], target, stats)
        if not ok:
            stats["fail"] += 1
            return None, lanes, stats
        total += cnt
        lanes.append((x0[l], end, cnt, ck))
    return total, lanes, stats
