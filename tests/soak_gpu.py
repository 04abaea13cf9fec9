"""Randomised parity soak on the GPU box (not collected by pytest):
    python tests/soak_gpu.py <first seed> <end seed>
General encoder (level 1 / RLE) and the PNG kernels against the oracle on random shapes and
contents, with guard bytes; several images per wavefront for the PNG pipeline."""
import os, sys, zlib
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import fdeflate_amd as fd
import oracle_binding as ob
import streams

def enc_round(seed):
    r = np.random.default_rng(seed)
    raws = []
    for k in range(96):
        kind = int(r.integers(0, 6))
        n = int(r.integers(0, 200000 if k % 17 == 0 else 20000))
        if kind == 0: a = r.integers(0, 256, n, dtype=np.uint8)
        elif kind == 1: a = r.integers(0, int(r.integers(1, 8)), n, dtype=np.uint8)
        elif kind == 2: a = np.repeat(r.integers(0, 256, n // 7 + 1, dtype=np.uint8), 7)[:n]
        elif kind == 3:
            base = r.integers(0, 256, max(1, n // 50), dtype=np.uint8); a = np.tile(base, 60)[:n]
        elif kind == 4:
            a = r.integers(0, 256, n, dtype=np.uint8); a[r.random(n) < 0.8] = 0
        else:
            a = (np.cumsum(r.integers(-2, 3, n)) & 0xFF).astype(np.uint8)
        raws.append(a.tobytes())
    buf, in_off = streams.pack_exact(raws)
    caps = [fd.compress_bound(len(x)) + 3 for x in raws]
    out_off = np.zeros(len(raws) + 1, dtype=np.int64); out_off[1:] = np.cumsum(caps)
    d_in = torch.from_numpy(buf).cuda(); d_io = torch.from_numpy(in_off.astype(np.int64)).cuda(); d_oo = torch.from_numpy(out_off).cuda()
    for mode, enc in ((fd.MODE_LEVEL1, ob.compress_level1), (fd.MODE_RLE, ob.compress_rle)):
        d_out = torch.full((int(out_off[-1]),), 0x5A, dtype=torch.uint8, device="cuda")
        ln = fd.deflate_general_batch(d_in, d_io, d_out, d_oo, mode).cpu().numpy().view(np.uint32)
        h = d_out.cpu().numpy()
        for i, raw in enumerate(raws):
            exp = enc(raw)
            got = h[out_off[i]:out_off[i] + int(ln[i])].tobytes()
            assert got == exp, (seed, mode, i, len(raw))
            assert np.all(h[out_off[i] + int(ln[i]):out_off[i + 1]] == 0x5A)

def png_round(seed):
    r = np.random.default_rng(seed)
    bpp = int(r.choice([1, 2, 3, 4, 6, 8]))
    row_bytes = bpp * int(r.integers(1, 400))
    pixs, types = [], []
    for k in range(int(r.integers(1, 60))):
        rows = int(r.integers(0, 150))
        pixs.append(bytes(r.integers(0, 256 if k % 3 else 3, row_bytes * rows, dtype=np.uint8)))
        types.append(bytes(r.integers(0, 5, rows, dtype=np.uint8)))
    filts = [ob.png_filter(p, row_bytes, bpp, t)[1] for p, t in zip(pixs, types)]
    pbuf, poff = streams.pack_exact(pixs); tbuf, toff = streams.pack_exact(types); fbuf, foff = streams.pack_exact(filts)
    d = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    for pw in ("1", "2", "5", "8"):
        os.environ["FDH_PNG_IMAGES_PER_WAVE"] = pw
        d_f = torch.full((int(foff[-1]) + 32,), 0xEE, dtype=torch.uint8, device="cuda")
        st = fd.png_filter_batch(d(pbuf), d(poff.astype(np.int64)), d(tbuf), d(toff.astype(np.int64)), d_f, d(foff.astype(np.int64)), row_bytes, bpp)
        h = d_f.cpu().numpy(); nf = int(foff[-1])
        assert int(st.abs().sum()) == 0 and h[:nf].tobytes() == fbuf[:nf].tobytes() and np.all(h[nf:] == 0xEE), (seed, pw, "filter")
        d_o = torch.full((int(poff[-1]) + 32,), 0xEE, dtype=torch.uint8, device="cuda")
        st = fd.png_unfilter_batch(d(fbuf), d(foff.astype(np.int64)), d_o, d(poff.astype(np.int64)), row_bytes, bpp)
        h = d_o.cpu().numpy(); npx = int(poff[-1])
        assert int(st.abs().sum()) == 0 and h[:npx].tobytes() == pbuf[:npx].tobytes() and np.all(h[npx:] == 0xEE), (seed, pw, "unfilter", bpp, row_bytes)

for s in range(int(sys.argv[1]), int(sys.argv[2])):
    enc_round(1000 + s); png_round(2000 + s)
    print("seed", s, "ok", flush=True)
print("SOAK OK")
