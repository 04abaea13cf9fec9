"""Executable CPU model of the device's speculative tile decoder (fdeflate_amd/csrc/
inflate_stream.h, Inflater::tile_step): 64 lanes x 64 bits, per-lane chains from guessed starts,
lane-to-lane synchronisation, prefix-summed output offsets, in-order match replay.

It models the ALGORITHM (lane by lane, in lock-step rounds), not the instruction stream, and is
used by tests/test_tile_model.py to show the algorithm reproduces zlib's output; the GPU parity
tests then pin the HIP implementation itself.
"""
import numpy as np

LIT_BITS, DIST_BITS = 12, 9
K_LIT1, K_LIT2, K_LEN, K_EOB, K_LONG = range(5)
LEN_BASE = [3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115,
            131, 163, 195, 227, 258]
LEN_EXTRA = [0] * 8 + [1] * 4 + [2] * 4 + [3] * 4 + [4] * 4 + [5] * 4 + [0]
DIST_BASE = [1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537,
             2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577]
DIST_EXTRA = [0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13]


def _codes(lengths):
    out, code = {}, 0
    for ln in range(1, 16):
        for s, l in enumerate(lengths):
            if l == ln:
                out[s] = (int(format(code, "0%db" % ln)[::-1], 2), ln)
                code += 1
        code <<= 1
    return out


def build_tables(litlen_lengths, dist_lengths):
    """Device-layout tables as tuples: lit[idx] = (kind, nbits, n1, s1, s2, base, extra)."""
    lit = [None] * (1 << LIT_BITS)
    for s, (c, l) in _codes(litlen_lengths).items():
        if l > LIT_BITS:
            lit[c & ((1 << LIT_BITS) - 1)] = (K_LONG, 0, 0, 0, 0, 0, 0)
            continue
        if s < 256:
            e = (K_LIT1, l, l, s, 0, 0, 0)
        elif s == 256 or s >= 286:
            e = (K_EOB, l, l, 0, 0, 0, 0)
        else:
            e = (K_LEN, l, l, 0, 0, LEN_BASE[s - 257], LEN_EXTRA[s - 257])
        for idx in range(c, 1 << LIT_BITS, 1 << l):
            lit[idx] = e
    single = list(lit)
    for idx in range(1 << LIT_BITS):
        e1 = single[idx]
        if e1[0] != K_LIT1:
            continue
        e2 = single[idx >> e1[1]]
        if e2[0] == K_LIT1 and e1[1] + e2[1] <= LIT_BITS:
            lit[idx] = (K_LIT2, e1[1] + e2[1], e1[1], e1[3], e2[3], 0, 0)
    dist = [("invalid", 0, 0, 0)] * (1 << DIST_BITS)
    nz = [i for i, l in enumerate(dist_lengths) if l]
    if len(nz) == 1 and dist_lengths[nz[0]] == 1:
        for idx in range(0, 1 << DIST_BITS, 2):
            dist[idx] = ("dist", 1, DIST_BASE[nz[0]], DIST_EXTRA[nz[0]]) if nz[0] < 30 else ("invalid", 1, 0, 0)
    else:
        for s, (c, l) in _codes(dist_lengths).items():
            if l > DIST_BITS:
                dist[c & ((1 << DIST_BITS) - 1)] = ("long", 0, 0, 0)
                continue
            e = ("dist", l, DIST_BASE[s], DIST_EXTRA[s]) if s < 30 else ("invalid", l, 0, 0)
            for idx in range(c, 1 << DIST_BITS, 1 << l):
                dist[idx] = e
    return lit, dist


class TileModel:
    def __init__(self, data, lit, dist):
        self.bits = int.from_bytes(data, "little")
        self.nbits = len(data) * 8
        self.lit, self.dist = lit, dist
        self.out = bytearray()

    def peek(self, pos, n):
        return (self.bits >> pos) & ((1 << n) - 1)

    def token(self, base, p, la):
        """-> (kind, adv1, adv): kind 0 literal(s), 1 match, 2 eob, 3 bad"""
        e = self.lit[self.peek(base + p, LIT_BITS)]
        k, nb = e[0], e[1]
        if k in (K_LIT1, K_LIT2):
            kind, a1, a = 0, e[2], nb
        elif k == K_LEN:
            t = nb + e[6]
            de = self.dist[self.peek(base + p + t, DIST_BITS)]
            kind = 1 if de[0] == "dist" else 3
            a = t + de[1] + de[3]
            a1 = a
        elif k == K_EOB:
            kind, a1, a = 2, nb, nb
        else:
            kind, a1, a = 3, 0, 0
        if p + a > la:
            kind = 3
        return kind, a1, a

    def tile(self, P):
        """One tile starting at stream bit P.  -> (bits used, reached end-of-block, bad)"""
        W, S = 64, 64
        left = self.nbits - P
        lanes = []
        for i in range(W):
            base, la = P + S * i, left - S * i
            L = dict(base=base, la=la, mask=0, mmask=0, start=0, stop=0, stop_pos=0, stop_nb=0, end=0)
            p = 0
            while p < S:
                kind, a1, a = self.token(base, p, la)
                if kind == 0:
                    L["mask"] |= 1 << p
                    if a != a1 and p + a1 < S:
                        L["mask"] |= 1 << (p + a1)
                        p += a
                    else:
                        p += a1
                elif kind == 1:
                    L["mask"] |= 1 << p
                    L["mmask"] |= 1 << p
                    p += a
                else:
                    L["stop"], L["stop_pos"], L["stop_nb"] = (1 if kind == 2 else 2), p, a
                    break
            L["end"] = p
            lanes.append(L)
        converged = False
        for _ in range(2 * W):
            prev_end = [lanes[max(i - 1, 0)]["end"] for i in range(W)]
            prev_stop = [lanes[max(i - 1, 0)]["stop"] for i in range(W)]
            in_start = [prev_end[i] - S for i in range(W)]
            need = [i != 0 and prev_stop[i] == 0 and in_start[i] != lanes[i]["start"] for i in range(W)]
            if not any(need):
                converged = True
                break
            for i in range(W):
                if not need[i]:
                    continue
                L = lanes[i]
                nm = nmm = 0
                p = in_start[i]
                while True:
                    if p >= S:
                        L.update(mask=nm, mmask=nmm, stop=0, end=p)
                        break
                    if (L["mask"] >> p) & 1:
                        keep = ~((1 << p) - 1)
                        L["mask"] = nm | (L["mask"] & keep)
                        L["mmask"] = nmm | (L["mmask"] & keep)
                        break
                    kind, a1, a = self.token(L["base"], p, L["la"])
                    if kind <= 1:
                        nm |= 1 << p
                        if kind == 1:
                            nmm |= 1 << p
                        p += a1
                    else:
                        L.update(mask=nm, mmask=nmm, stop=(1 if kind == 2 else 2), stop_pos=p, stop_nb=a, end=p)
                        break
                L["start"] = in_start[i]
        assert converged
        stop_lane = W
        for i in range(W):
            if lanes[i]["stop"]:
                stop_lane = i
                break
        used, eob, bad = None, False, False
        for i in range(W):
            if i > stop_lane:
                break
            L = lanes[i]
            m = L["mask"]
            p = 0
            while m:
                p = (m & -m).bit_length() - 1
                m &= m - 1
                e = self.lit[self.peek(L["base"] + p, LIT_BITS)]
                if e[0] in (K_LIT1, K_LIT2):
                    self.out.append(e[3])
                    if e[0] == K_LIT2 and p + e[2] < S:
                        self.out.append(e[4])
                        m &= ~(1 << (p + e[2]))
                else:
                    lcb, lex = e[1], e[6]
                    length = e[5] + self.peek(L["base"] + p + lcb, lex)
                    de = self.dist[self.peek(L["base"] + p + lcb + lex, DIST_BITS)]
                    d = de[2] + self.peek(L["base"] + p + lcb + lex + de[1], de[3])
                    assert d <= len(self.out)
                    for _ in range(length):
                        self.out.append(self.out[-d])
        if stop_lane < W:
            L = lanes[stop_lane]
            used = stop_lane * S + L["stop_pos"]
            if L["stop"] == 1:
                used += L["stop_nb"]
                eob = True
            else:
                bad = True
        else:
            used = (W - 1) * S + lanes[W - 1]["end"]
        return used, eob, bad
