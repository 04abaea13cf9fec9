#!/usr/bin/env python3
"""Diagnostics of the interval kernel (inflate_seg2.h) on the bench workload (GPU box):
    python tools/seg2diag.py [n_streams] [model] [length]
  * how many streams inflate_seg2_kernel finishes itself (flag 0x800: that kernel only) and whether
    what it wrote is right
  * its time alone vs the whole pipeline vs the pipeline without it (flag 0x400)"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import fdeflate_amd as fd  # noqa: E402
from fdeflate_amd import synth  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
model = sys.argv[2] if len(sys.argv) > 2 else "D"
L = int(sys.argv[3]) if len(sys.argv) > 3 else 65536
dev = "cuda"
raw = synth.gen_batch_torch(0, n, L, model=model, device=dev)
r_off = torch.arange(n + 1, dtype=torch.int64, device=dev) * L
bound = (fd.ultrafast_bound(L) + 15) & ~15
t_off = torch.arange(n + 1, dtype=torch.int64, device=dev) * bound
tmp = torch.zeros(n * bound, dtype=torch.uint8, device=dev)
clen = fd.deflate_ultrafast_batch(raw.view(-1), r_off, tmp, t_off)
del tmp
t_off = torch.zeros(n + 1, dtype=torch.int64, device=dev)
t_off[1:] = torch.cumsum((clen.to(torch.int64) + 15) & ~15, 0)
comp = torch.zeros(int(t_off[-1]), dtype=torch.uint8, device=dev)
fd.deflate_ultrafast_batch(raw.view(-1), r_off, comp, t_off)
out = torch.empty(n * L, dtype=torch.uint8, device=dev)
ol = torch.empty(n, dtype=torch.int32, device=dev)
st = torch.empty(n, dtype=torch.int32, device=dev)
ad = torch.empty(n, dtype=torch.int32, device=dev)


def timed(flags, reps=10, trials=3):
    fd.inflate_batch(comp, t_off, out, r_off, ol, st, ad, flags=flags)
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(trials):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fd.inflate_batch(comp, t_off, out, r_off, ol, st, ad, flags=flags)
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / reps)
    return best


out.zero_()
st.fill_(7)
fd.inflate_batch(comp, t_off, out, r_off, ol, st, ad, flags=0x800)
torch.cuda.synchronize()
done = st == 0
rows_out = out.view(n, L)
rows_raw = raw.view(n, L)
eq = (rows_out == rows_raw).all(dim=1)
wrong = done & ~eq
print("streams %d model %s length %d mean compressed %.0f B" % (n, model, L, float(clen.sum()) / n))
print("interval kernel alone: finished %d of %d itself, %d of them WRONG, lengths ok: %s"
      % (int(done.sum()), n, int(wrong.sum()), bool((ol[done] == L).all())))
if int(wrong.sum()):
    for i in torch.nonzero(wrong).flatten()[:4].tolist():
        d = torch.nonzero(rows_out[i] != rows_raw[i]).flatten()
        print("  stream %d: %d bytes differ, first at %d: got %s want %s" % (
            i, d.numel(), int(d[0]), rows_out[i][int(d[0]):int(d[0]) + 8].tolist(), rows_raw[i][int(d[0]):int(d[0]) + 8].tolist()))
left = torch.nonzero(~done).flatten()
if left.numel():
    kinds = {}
    for i in left.tolist():
        kinds[i % 16] = kinds.get(i % 16, 0) + 1
    print("  left over by stream index mod 16:", dict(sorted(kinds.items())), "first:", left[:8].tolist())
t_new = timed(0x800)
t_all = timed(0)
ok_all = bool(torch.equal(out, raw.view(-1))) and bool((st == 0).all())
t_old = timed(0x400)
print("interval kernel alone %.3f ms | whole pipeline %.3f ms (all right: %s) | without it %.3f ms"
      % (t_new, t_all, ok_all, t_old))
print("=> scaled to 65536 streams: %.2f ms (was %.2f)" % (t_all * 65536 / n, t_old * 65536 / n))

# instrumented build (FDH_LIB=.../libfdeflate_hip_s2dbg.so): records of given streams
from fdeflate_amd import _lib  # noqa: E402
import ctypes as C  # noqa: E402
import numpy as np  # noqa: E402
Lc = _lib.lib()
if hasattr(Lc, "fdh_debug_s2"):
    targets = [int(x) for x in os.environ.get("FDH_S2_SIDS", "").split(",") if x]
    if not targets:
        targets = torch.nonzero(wrong).flatten()[:2].tolist() + left[:2].tolist()
    buf = np.zeros(8 * 2048, dtype=np.uint32)
    for sid in targets:
        Lc.fdh_debug_s2(buf.ctypes.data_as(C.c_void_p), C.c_uint32(sid))
        fd.inflate_batch(comp, t_off, out, r_off, ol, st, ad, flags=0x800)
        torch.cuda.synchronize()
        nrec = Lc.fdh_debug_s2(buf.ctypes.data_as(C.c_void_p), C.c_uint32(0xFFFFFFFF))
        print("---- stream %d (kind %d): %d records, status %d" % (sid, sid % 16, nrec, int(st[sid])))
        for r in buf[: 8 * min(nrec, 2048)].reshape(-1, 8):
            tag = int(r[0])
            v = [int(x) for x in r[1:]]
            if tag == 4:
                print("  plan: total %d ni %d seg %d tb %d stop %08x ok %d bulk lines %d" % tuple(v))
            elif tag == 1:
                print("  round f0 %d n %d wq %d qa %d qf_new %d qa_new %d n_brk %d bla %d" % (v[0], v[1], (v[2] - (1 << 32) if v[2] >> 31 else v[2]), v[3], v[4], v[5], v[6] & 255, v[6] >> 8))
            elif tag == 2:
                print("    chain xi %d len %d bl_front %d v %d kl %d byte %d wq %d" % (v[0], v[1], v[2], v[3], v[4], v[5], (v[6] - (1 << 32) if v[6] >> 31 else v[6])))
            elif tag == 3:
                if v[2] or v[3] or v[4]:
                    print("  BAD f0 %d n %d bad mask %08x%08x short mask %08x" % (v[0], v[1], v[3], v[2], v[4]))
            elif tag == 5:
                print("  adler %08x total %d" % (v[0], v[1]))
        d = torch.nonzero(rows_out[sid] != rows_raw[sid]).flatten()
        if d.numel():
            print("  differing bytes: %d, first %d last %d" % (d.numel(), int(d[0]), int(d[-1])))
if hasattr(Lc, "fdh_debug_s2time"):
    fd.inflate_batch(comp, t_off, out, r_off, ol, st, ad, flags=0x800)
    torch.cuda.synchronize()
    tb = np.zeros(4096 * 24, dtype=np.uint32)
    Lc.fdh_debug_s2time(tb.ctypes.data_as(C.c_void_p))
    t = tb[:4096 * 16].reshape(4096, 16).astype(np.int64)
    t2 = tb[4096 * 16:].reshape(4096, 8).astype(np.int64)
    m = min(n, 4096)
    sel = np.array([i for i in range(m) if i % 16 not in (7, 15)])
    d = (t[:, 1:6] - t[:, 0:5]) & 0xFFFFFFFF
    names = ["guess", "tail count", "check", "plan", "write pass"]
    tot = d[sel].sum(axis=1).mean()
    print("cycles per noisy stream (lane 0 of its wavefront, clock64): total %.0f" % tot)
    for k, nm in enumerate(names):
        print("  %-12s %8.0f  %5.1f %%" % (nm, d[sel, k].mean(), 100 * d[sel, k].mean() / tot))
    wn = ["round start", "next round: intervals, fit, input request", "input image + lane set-up", "groups", "chains", "checks + waiting runs", "flush", "carry"]
    acc = t[:, 8:16]
    wtot = acc[sel].sum(axis=1).mean()
    for k, nm in enumerate(wn):
        if nm:
            v = acc[sel, k].mean()
            print("    write: %-40s %8.0f  %5.1f %%" % (nm, v, 100 * v / wtot))
    tn = ["refill at start", "events", "predicates + meter", "group", "general step", "", "", "(entry)"]
    for k, nm in enumerate(tn):
        if nm:
            print("    tail: %-30s %8.0f" % (nm, t2[sel, k].mean()))
    for kind in (7, 15):
        selk = np.array([i for i in range(m) if i % 16 == kind])
        print("  kind %d streams (%s): total %.0f cycles: %s" % (
            kind, "half zero" if kind == 7 else "all zero", d[selk].sum(axis=1).mean(),
            ", ".join("%s %.0f" % (nm, d[selk, k].mean()) for k, nm in enumerate(names))))
        print("      write pass: " + ", ".join("%s %.0f" % (nm.split(":")[0][:18], acc[selk, k].mean()) for k, nm in enumerate(wn)))
        print("      tail scan:  " + ", ".join("%s %.0f" % (nm, t2[selk, k].mean()) for k, nm in enumerate(tn) if nm))
