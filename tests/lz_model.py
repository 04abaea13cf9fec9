"""Executable CPU model of the LZ-window decoder (fdeflate_amd/csrc/inflate_lz.h): the algorithm of one
"span" of a compressed block, lane by lane in lock-step, as the HIP kernel runs it.

  * the next 64 x R stream bits are cut into 64 ranges, one per lane; a lane walks a guessed chain from up
    to W bits in front of its range (impossible tokens slide on by one bit) and counts the output bytes of
    the tokens that start inside its range;
  * check: a lane's first token start at or behind its range start must be where its left neighbour's
    chain left the neighbour's range (induction from lane 0, whose start is real); lanes that fail walk
    again from the neighbour's end;
  * a wave prefix sum gives output offsets; as many lanes as fit the image take part;
  * pass 2 decodes again: literals go to the image, a match leaves a 3-byte descriptor
    (length - 3, distance - 1) in its first three bytes and a bit in the start bitmap;
  * matches are resolved by output position, 64 positions at a time (lane = position): the covering
    match comes from the bitmap (nearest start at or below) or is carried over from the group before;
    a byte is out[start - dist + (k mod dist)], which lies in front of the match, so a byte waits only
    for bytes of EARLIER tokens of its own group.

It models the algorithm, not the instruction stream; tests/test_lz_model.py checks it against zlib, the
GPU parity tests pin the HIP implementation.  Reference semantics: src/decompress.rs:611-1018.
"""
import zlib

LEN_BASE = [3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115,
            131, 163, 195, 227, 258]
LEN_EXTRA = [0] * 8 + [1] * 4 + [2] * 4 + [3] * 4 + [4] * 4 + [5] * 4 + [0]
DIST_BASE = [1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537,
             2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577]
DIST_EXTRA = [0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13]
CL_ORDER = [16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15]
LANES = 64


class Bail(Exception):
    """anything the fast kernel leaves to the exact kernels behind it"""


def _table(lengths):
    maxl = max(lengths) if lengths else 0
    if maxl == 0:
        return [], 0
    tab = [None] * (1 << maxl)
    code = 0
    for ln in range(1, maxl + 1):
        for s, l in enumerate(lengths):
            if l == ln:
                rev = int(format(code, "0%db" % ln)[::-1], 2)
                for idx in range(rev, 1 << maxl, 1 << ln):
                    tab[idx] = (s, ln)
                code += 1
        code <<= 1
    return tab, maxl


class Stream:
    def __init__(self, data):
        self.v = int.from_bytes(data, "little")
        self.nbits = len(data) * 8

    def peek(self, pos, k):
        return (self.v >> pos) & ((1 << k) - 1)


class LzModel:
    def __init__(self, data, R=544, W=512, img_cap=8192, pairs=True):
        self.st = Stream(data)
        self.R, self.W, self.img_cap, self.pairs = R, W, img_cap, pairs
        self.out = bytearray()
        self.stats = dict(spans=0, fixups=0, groups=0, iters=0)

    # ---- one token at bit `pos`: (kind, bits, n_out, payload, bits_first) ----
    # kind 0 literal(s) (payload = bytes), 1 match (payload = (length, dist)), 2 end-of-block, 3 impossible
    def token(self, pos, single):
        st = self.st
        e = self.LT[st.peek(pos, self.lm)] if self.lm else None
        if e is None:
            return (3, 0, 0, None)
        s, l = e
        if s < 256:
            if self.pairs and not single:  # a second literal whose code fits the 10 index bits with the first
                e2 = self.LT[st.peek(pos + l, self.lm)]
                if e2 is not None and e2[0] < 256 and l + e2[1] <= 10:
                    return (0, l + e2[1], 2, bytes([s, e2[0]]), l)
            return (0, l, 1, bytes([s]), l)
        if s == 256 or s >= 286:
            return (2, l, 0, None, l)
        i = s - 257
        length = LEN_BASE[i] + st.peek(pos + l, LEN_EXTRA[i])
        t = l + LEN_EXTRA[i]
        if not self.dm:
            return (3, 0, 0, None, 0)
        de = self.DT[st.peek(pos + t, self.dm)]
        if de is None or de[0] >= 30:
            return (3, 0, 0, None, 0)
        ds, dl = de
        dist = DIST_BASE[ds] + st.peek(pos + t + dl, DIST_EXTRA[ds])
        bits = t + dl + DIST_EXTRA[ds]
        return (1, bits, length, (length, dist), bits)

    # ---- a lane's walk: from `start` (a real boundary if `real`) to the first token start >= end ----
    # returns (first token start >= s, end position, bytes counted, stop) ; stop: 0 none, 1 end-of-block, 2 bad
    def walk(self, start, s, end, real, emit=None):
        pos, cnt, b, stop = start, 0, None, 0
        while pos < end:
            started = pos >= s
            if started and b is None:
                b = pos
            mark = end if started else s
            tk = self.token(pos, False)
            if tk[0] == 0 and tk[2] == 2 and pos + tk[4] >= mark:  # the second literal belongs to the next range
                tk = self.token(pos, True)
            kind, bits = tk[0], tk[1]
            if pos + bits > self.st.nbits and kind != 3:
                kind = 3
            if kind >= 2:
                if not started and not real:
                    pos += 1  # guessed chain in front of its range: slide on
                    continue
                # (a real chain that stops in front of its range: the block ends in a lane to the left, which
                #  sees the same stop inside its own range; this lane is not part of the span)
                stop = 1 if kind == 2 else 2
                break
            if started:
                cnt += tk[2]
                if emit:
                    emit(tk)
            pos += bits
        if b is None:
            b = pos
        return b, pos, cnt, stop

    def span(self, S, R):
        st = self.st
        self.stats["spans"] += 1
        O = len(self.out)
        s = [S + l * R for l in range(LANES)]
        end = [x + R for x in s]
        live = [x < st.nbits for x in s]
        b, e, cnt, stop = [0] * LANES, [0] * LANES, [0] * LANES, [0] * LANES
        for l in range(LANES):
            if not live[l]:
                b[l] = e[l] = s[l]
                stop[l] = 2
                continue
            ws = max(S, s[l] - self.W)
            b[l], e[l], cnt[l], stop[l] = self.walk(ws, s[l], end[l], ws == S)
        # ---- check + fix-up rounds ----
        start = list(b)
        for rnd in range(LANES + 1):
            ok = [True] * LANES
            for l in range(1, LANES):
                ok[l] = stop[l - 1] == 0 and start[l] == e[l - 1]
            # lanes behind the first verified stop are not part of this span
            first_bad = next((l for l in range(LANES) if not ok[l]), LANES)
            first_stop = next((l for l in range(first_bad) if stop[l]), None)
            if first_stop is not None or first_bad == LANES:
                break
            self.stats["fixups"] += 1
            redo = [l for l in range(1, LANES) if not ok[l] and stop[l - 1] == 0]
            newv = {}
            for l in redo:
                st_l = e[l - 1]
                if st_l >= end[l]:  # the neighbour's last token covers this whole range
                    newv[l] = (st_l, st_l, 0, 0)
                else:
                    newv[l] = self.walk(st_l, s[l], end[l], True)
                    newv[l] = (st_l,) + newv[l][1:]
            for l, v in newv.items():
                start[l], e[l], cnt[l], stop[l] = v
        else:
            raise Bail("no convergence")
        nvalid = LANES if first_stop is None else first_stop + 1
        if first_stop is not None and stop[first_stop] == 2:
            raise Bail("bad token on the real chain")
        # ---- offsets, as many lanes as fit the image ----
        pre = [0] * (LANES + 1)
        for l in range(nvalid):
            pre[l + 1] = pre[l] + cnt[l]
        nuse = nvalid
        while nuse > 0 and pre[nuse] > self.img_cap:
            nuse -= 1
        if nuse == 0:
            return None  # caller retries with a smaller R
        N = pre[nuse]
        # ---- pass 2 ----
        img = bytearray(N)
        bitmap = bytearray(N)
        for l in range(nuse):
            q = [pre[l]]

            def emit(tk, q=q):
                if tk[0] == 0:
                    for byte in tk[3]:
                        img[q[0]] = byte
                        q[0] += 1
                else:
                    length, dist = tk[3]
                    if dist > O + q[0]:
                        raise Bail("distance too far back")
                    d = (length - 3) | ((dist - 1) << 8)
                    img[q[0]] = d & 0xFF
                    img[q[0] + 1] = (d >> 8) & 0xFF
                    img[q[0] + 2] = (d >> 16) & 0xFF
                    bitmap[q[0]] = 1
                    q[0] += length

            if start[l] < end[l]:
                b2, e2, c2, s2 = self.walk(start[l], start[l], end[l], True, emit)
                assert (e2, c2, s2) == (e[l], cnt[l], stop[l]), "pass 2 disagrees with pass 1"
            assert q[0] == pre[l + 1]
        # ---- resolution: 64 output positions at a time ----
        self.out += img  # history + image addressed by absolute position
        out = self.out
        p0 = O & ~63
        carry = None  # (start position, descriptor)
        for p in range(p0, O + N, 64):
            self.stats["groups"] += 1
            cov, src = [False] * 64, [0] * 64
            last = carry
            descs = {}
            for j in range(64):  # every lane reads its descriptor before any lane writes
                q = p + j
                if O <= q < O + N and bitmap[q - O]:
                    descs[j] = out[q] | (out[q + 1] << 8) | (out[q + 2] << 16)
            for j in range(64):
                q = p + j
                if j in descs:
                    last = (q, descs[j])
                if last is None or not (O <= q < O + N):
                    continue
                spos, d = last
                length, dist = (d & 0xFF) + 3, (d >> 8) + 1
                k = q - spos
                if k < length:
                    cov[j] = True
                    src[j] = spos - dist + (k % dist)
            carry = last
            pend = [j for j in range(64) if cov[j]]
            while pend:
                self.stats["iters"] += 1
                done = [j not in pend for j in range(64)]
                ready = [j for j in pend if src[j] < p or done[src[j] - p]]
                assert ready, "dependency cycle"
                vals = {j: out[src[j]] for j in ready}
                for j, v in vals.items():
                    out[p + j] = v
                pend = [j for j in pend if j not in vals]
        # ---- where the stream continues ----
        last = nuse - 1
        if nuse == nvalid and first_stop is not None:
            tk = self.token(e[last], False)
            return e[last] + tk[1], True
        return e[last], False

    def parse_block_header(self, pos):
        st = self.st
        final, typ = st.peek(pos, 1), st.peek(pos + 1, 2)
        pos += 3
        if typ == 0 or typ == 3:
            raise Bail("stored / invalid block type")
        if typ == 1:
            lens = [8] * 144 + [9] * 112 + [7] * 24 + [8] * 8
            dl = [5] * 32
            if st.peek(pos, 7) == 0:
                return pos + 7, final, True
        else:
            hlit, hdist, hclen = st.peek(pos, 5) + 257, st.peek(pos + 5, 5) + 1, st.peek(pos + 10, 4) + 4
            pos += 14
            cl = [0] * 19
            for i in range(hclen):
                cl[CL_ORDER[i]] = st.peek(pos, 3)
                pos += 3
            CT, cm = _table(cl)
            lens = []
            while len(lens) < hlit + hdist:
                sy, l = CT[st.peek(pos, cm)]
                pos += l
                if sy < 16:
                    lens.append(sy)
                elif sy == 16:
                    lens += [lens[-1]] * (3 + st.peek(pos, 2))
                    pos += 2
                elif sy == 17:
                    lens += [0] * (3 + st.peek(pos, 3))
                    pos += 3
                else:
                    lens += [0] * (11 + st.peek(pos, 7))
                    pos += 7
            lens, dl = lens[:hlit], lens[hlit:]
        self.LT, self.lm = _table(lens)
        self.DT, self.dm = _table(dl)
        return pos, final, False

    def run(self):
        st = self.st
        pos = 16
        while True:
            pos, final, empty = self.parse_block_header(pos)
            eob = empty
            R = self.R
            while not eob:
                rem = st.nbits - pos
                r = min(R, max(32, -(-rem // LANES)))
                res = self.span(pos, r)
                if res is None:
                    if R <= 1:
                        raise Bail("a single token overflows the image")
                    R = max(1, R // 4)
                    continue
                pos, eob = res
            if final:
                break
        pos = (pos + 7) & ~7
        if pos + 32 > st.nbits:
            raise Bail("truncated trailer")
        stored = int.from_bytes(st.peek(pos, 32).to_bytes(4, "little"), "big")
        if stored != zlib.adler32(bytes(self.out)):
            raise Bail("checksum")
        return bytes(self.out)
