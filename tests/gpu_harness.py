"""Helpers for the -m gpu parity tests: pack streams, run the HIP path through the C ABI
(fdeflate_amd.inflate_batch / deflate_ultrafast_batch) and the oracle on the same bytes."""
import numpy as np

import oracle_binding as ob
import streams


def _t(a, dtype=None):
    import torch
    t = torch.from_numpy(np.ascontiguousarray(a))
    return t.cuda()


def gpu_inflate(blobs, caps, flags=0, guard=5, fill=0xA5):
    """Decodes blobs[i] into a slot of caps[i] bytes.  Slots are interleaved with `guard`-byte
    dummy slots (empty input) so every slot starts at an odd alignment and any out-of-slot write
    is detected.  -> (status, out_len, adler, outputs[list of bytes], guards_ok)"""
    import torch
    import fdeflate_amd as fd

    n = len(blobs)
    all_blobs, all_caps = [], []
    for b, c in zip(blobs, caps):
        all_blobs += [b, b""]
        all_caps += [c, guard]
    buf, in_off = streams.pack_exact(all_blobs)
    out_off = np.zeros(2 * n + 1, dtype=np.uint64)
    out_off[1:] = np.cumsum(np.asarray(all_caps, dtype=np.uint64))
    total = int(out_off[-1])
    d_in = _t(buf)
    d_in_off = _t(in_off.astype(np.int64))
    d_out = torch.full((max(total, 1),), fill, dtype=torch.uint8, device="cuda")
    d_out_off = _t(out_off.astype(np.int64))
    out_len, status, adler = fd.inflate_batch(d_in, d_in_off, d_out, d_out_off, flags=flags)
    torch.cuda.synchronize()
    h_out = d_out.cpu().numpy()
    st = status.cpu().numpy().view(np.uint32)[0::2]
    ln = out_len.cpu().numpy().view(np.uint32)[0::2]
    ad = adler.cpu().numpy().view(np.uint32)[0::2]
    outs, guards_ok = [], True
    for i in range(n):
        o0, o1, g1 = int(out_off[2 * i]), int(out_off[2 * i + 1]), int(out_off[2 * i + 2])
        outs.append(h_out[o0:o1])
        if not np.all(h_out[o1:g1] == fill):
            guards_ok = False
    return st, ln, ad, outs, guards_ok


def oracle_inflate(blobs, caps, ignore_adler32=False):
    sts, lens, ads, outs = [], [], [], []
    for b, c in zip(blobs, caps):
        st, out, ad = ob.decompress_bounded(b, c, ignore_adler32)
        sts.append(st)
        lens.append(len(out))
        ads.append(ad)
        outs.append(out)
    return sts, lens, ads, outs


def assert_inflate_parity(names, blobs, caps, flags=0):
    """Bit-exact: status for every stream; length, bytes and Adler-32 whenever the reference
    defines them (Ok and OutputTooLarge)."""
    st, ln, ad, outs, guards_ok = gpu_inflate(blobs, caps, flags)
    rs, rl, ra, ro = oracle_inflate(blobs, caps, bool(flags & 1))
    bad = []
    for i, name in enumerate(names):
        if int(st[i]) != rs[i]:
            bad.append((name, "status", ob.STATUS_NAMES[int(st[i])] if st[i] < 18 else int(st[i]),
                        ob.STATUS_NAMES[rs[i]], caps[i]))
            continue
        if rs[i] in (0, 17):
            if int(ln[i]) != rl[i]:
                bad.append((name, "len", int(ln[i]), rl[i], caps[i]))
            elif outs[i][:rl[i]].tobytes() != ro[i]:
                diff = np.nonzero(np.frombuffer(ro[i], dtype=np.uint8) != outs[i][:rl[i]])[0]
                bad.append((name, "bytes", "first diff at %d of %d" % (diff[0], rl[i]), caps[i]))
            elif rs[i] == 0 and int(ad[i]) != ra[i]:
                bad.append((name, "adler", hex(int(ad[i])), hex(ra[i])))
    assert guards_ok, "a kernel wrote outside its output slot"
    assert not bad, bad[:10]


def gpu_deflate(raws, guard=3, slack=0, fill=0x5A):
    import torch
    import fdeflate_amd as fd

    n = len(raws)
    all_raw, all_caps = [], []
    for r in raws:
        all_raw += [r, b""]
        all_caps += [fd.ultrafast_bound(len(r)) + slack, fd.ultrafast_bound(0) + guard]
    buf, in_off = streams.pack_exact(all_raw)
    out_off = np.zeros(2 * n + 1, dtype=np.uint64)
    out_off[1:] = np.cumsum(np.asarray(all_caps, dtype=np.uint64))
    d_in = _t(buf)
    d_out = torch.full((int(out_off[-1]),), fill, dtype=torch.uint8, device="cuda")
    out_len = fd.deflate_ultrafast_batch(d_in, _t(in_off.astype(np.int64)), d_out, _t(out_off.astype(np.int64)))
    torch.cuda.synchronize()
    h = d_out.cpu().numpy()
    ln = out_len.cpu().numpy().view(np.uint32)
    res = []
    ok = True
    for i in range(n):
        o0 = int(out_off[2 * i])
        res.append(h[o0:o0 + int(ln[2 * i])].tobytes())
        o1 = int(out_off[2 * i + 1])
        if not np.all(h[o0 + int(ln[2 * i]):o1] == fill):
            ok = False  # wrote past its own length inside the slot
    return res, ok
