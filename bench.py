#!/usr/bin/env python3
"""bench.py -- decompressed GB/s of the batched zlib decode (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W

One "step" = one pass of fdh_inflate_batch over the whole per-GPU batch (65 536 independent
64 KiB PNG-filter zlib streams in the ultra-fast format, BASELINE config 2), inputs resident in
HBM.  Streams are sharded by rank with no data-path collective; for N > 1 the per-stream metadata
is all-gathered (RCCL) inside the step and the step time is the max over ranks.

Launching: with N > 1 and no rank environment (WORLD_SIZE unset) this process only starts N rank
processes (one per GPU, free rendezvous port) and relays rank 0's JSON line -- it never touches
the GPU itself.  Under an external launcher (torch.distributed.run) every process is a rank.

Rank 0 prints ONE JSON line.  Besides the contract fields it carries
  roofline      the headline decode kernel against the HBM peak (live HIP-event time)
  cpu_baseline  the oracle (C port of the reference) and system zlib timed on the host cores
  also          (N = 1) the ultra-fast ENCODE of the same buffers (BASELINE config 3) and the
                decode of zlib level-6 streams of the same data (config 2 (ii)), each with its
                own roofline object
  payload_gather_ms  (N > 1) one RCCL all-gather of the decoded payload, timed separately
"""
import argparse
import hashlib
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: 8.0 TB/s spec


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--streams", type=int, default=65536, help="streams per GPU")
    ap.add_argument("--stream-bytes", type=int, default=65536)
    ap.add_argument("--mode", choices=["decode", "encode"], default="decode")
    ap.add_argument("--format", choices=["ultrafast", "zlib6"], default="ultrafast")
    ap.add_argument("--flags", type=int, default=0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-also", action="store_true", help="skip the extra encode / zlib-6 lines")
    ap.add_argument("--general-streams", type=int, default=65536, help="streams given to the level-1 / RLE encoders (also lines)")
    ap.add_argument("--cpu-seconds", type=float, default=5.0, help="per CPU-baseline leg (4 legs)")
    ap.add_argument("--zlib6-streams", type=int, default=65536, help="SURVEY 8(d) C2 (ii): the same buffers as the headline")
    ap.add_argument("--also-select", default="encode,level1,rle,png,zlib6,mix",
                    help="which extra lines to run (tools/profile.sh profiles them one by one)")
    ap.add_argument("--no-payload-gather", action="store_true")
    ap.add_argument("--force-dist", action="store_true",
                    help="testing: run the RCCL metadata gather even with one rank")
    return ap.parse_args()


# ------------------------------------------------------------------------------------------
# launcher (parent of the ranks; never initialises the GPU)
# ------------------------------------------------------------------------------------------

def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def launch_ranks(n):
    """Starts n copies of this script as ranks 0..n-1 and relays rank 0's output."""
    port = free_port()
    procs = []
    for r in range(n):
        env = dict(os.environ)
        env.update({"RANK": str(r), "LOCAL_RANK": str(r), "WORLD_SIZE": str(n), "LOCAL_WORLD_SIZE": str(n),
                    "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port)})
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        out = None if r == 0 else subprocess.DEVNULL
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=out))
    # wait for all of them; a rank that dies leaves the others waiting at a collective, so the first
    # failure ends the run
    rc = 0
    live = list(procs)
    while live and rc == 0:
        time.sleep(0.2)
        for p in list(live):
            code = p.poll()
            if code is not None:
                live.remove(p)
                if code != 0:
                    rc = code
    for p in live:
        p.kill()
    for p in live:
        p.wait()
    return rc


# ------------------------------------------------------------------------------------------
# inputs
# ------------------------------------------------------------------------------------------

def encode_ultrafast(raw, r_off, dev):
    """raw [n, L] on the device -> its ultra-fast encoding packed 16-B aligned."""
    import torch
    import fdeflate_amd as fd
    n, L = raw.shape
    # pass 1: lengths only (slots of the worst-case bound), pass 2: exact packed layout
    bound = (fd.ultrafast_bound(L) + 15) & ~15
    tmp = torch.empty(n * bound, dtype=torch.uint8, device=dev)
    t_off = torch.arange(n + 1, dtype=torch.int64, device=dev) * bound
    clen = fd.deflate_ultrafast_batch(raw.view(-1), r_off, tmp, t_off).to(torch.int64)
    del tmp
    padded = (clen + 15) & ~15
    c_off = torch.zeros(n + 1, dtype=torch.int64, device=dev)
    c_off[1:] = torch.cumsum(padded, 0)
    comp = torch.zeros(int(c_off[-1]), dtype=torch.uint8, device=dev)
    clen2 = fd.deflate_ultrafast_batch(raw.view(-1), r_off, comp, c_off).to(torch.int64)
    assert torch.equal(clen, clen2)
    return comp, c_off, clen


def _z6(b):
    import zlib
    return zlib.compress(b, 6)


def encode_zlib6(raw_rows, dev):
    """zlib level 6 (system zlib, host threads: zlib.compress releases the GIL) of the rows of a
    host array -> packed on the device."""
    from concurrent.futures import ThreadPoolExecutor
    import numpy as np
    import torch
    n = raw_rows.shape[0]
    rows = [raw_rows[i].tobytes() for i in range(n)]
    with ThreadPoolExecutor(max(1, min(64, (os.cpu_count() or 1)))) as pool:
        blobs = list(pool.map(_z6, rows, chunksize=64))
    clen_h = np.array([len(b) for b in blobs], dtype=np.int64)
    off_h = np.zeros(n + 1, dtype=np.int64)
    off_h[1:] = np.cumsum((clen_h + 15) & ~15)
    buf = np.zeros(int(off_h[-1]), dtype=np.uint8)
    for i, b in enumerate(blobs):
        buf[off_h[i]:off_h[i] + len(b)] = np.frombuffer(b, dtype=np.uint8)
    return torch.from_numpy(buf).to(dev), torch.from_numpy(off_h).to(dev), torch.from_numpy(clen_h).to(dev)


def mix_pool(raw, uf_comp, uf_off, uf_len, rnd):
    """The distinct streams of the mix line: (compressed, raw or None, Ok expected, what it is, status expected).
    The status of a stream that is not Ok is what its maker built it for (include/fdeflate_hip.h fdh_stream_status:
    1 BadZlibHeader, 2 InsufficientInput, 9 BadLiteralLengthHuffmanTree, 15 WrongChecksum); the three .zz vectors are
    the reference's own (src/decompress.rs:1344-1384: a wrong checksum, two trees without an end-of-block code)."""
    import zlib
    pool = []
    gold = os.path.join(ROOT, "tests", "golden", "vectors")
    nref = 0
    for name in sorted(os.listdir(os.path.join(gold, "corpus"))):
        c = open(os.path.join(gold, "corpus", name), "rb").read()
        pool.append((c, zlib.decompress(c), True, "corpus/" + name, 0))
        nref += 1
    for name in sorted(os.listdir(gold)):
        if name.endswith(".zz"):   # (a wrong checksum, two trees without an end-of-block code: reference tests/*.zz)
            c = open(os.path.join(gold, name), "rb").read()
            try:
                pool.append((c, zlib.decompress(c), True, name, 0))
            except zlib.error as ze:
                pool.append((c, None, False, name, 15 if "incorrect data check" in str(ze) else 9))
            nref += 1
    h = raw[:48].cpu().numpy()
    for i in range(48):
        r = h[i].tobytes()
        kind = i % 8
        if kind == 0:
            c = zlib.compress(r, 0)                                   # stored blocks
        elif kind in (1, 2, 3):
            o = zlib.compressobj(6, zlib.DEFLATED, 15, 9, (zlib.Z_FIXED, zlib.Z_RLE, zlib.Z_HUFFMAN_ONLY)[kind - 1])
            c = o.compress(r) + o.flush()
        elif kind == 4:
            c = zlib.compress(r, 9)
        elif kind == 5:
            r = r[:rnd.randrange(1, 9000)]
            c = zlib.compress(r, 6)
        else:
            c = zlib.compress(r, 6)
        pool.append((c, r, True, "zlib kind %d of buffer %d" % (kind, i), 0))
    ufo = uf_off[:33].cpu().numpy()
    ufl = uf_len[:32].cpu().numpy()
    ufc = uf_comp[:int(ufo[32])].cpu().numpy()
    for i in range(32):                                               # the headline's own format
        pool.append((ufc[int(ufo[i]):int(ufo[i]) + int(ufl[i])].tobytes(), h[i].tobytes(), True, "ultra-fast, buffer %d" % i, 0))
    good = [p for p in pool if p[2] and len(p[0]) > 64]
    for k in range(24):                                               # damaged and truncated copies: never Ok
        g = good[rnd.randrange(len(good))]
        c = g[0]
        if k % 3 == 0:
            pool.append((c[:rnd.randrange(8, len(c) - 4)], None, False, "cut short: " + g[3], 2))   # InsufficientInput
        elif k % 3 == 1:
            b = bytearray(c)
            b[len(b) - 1 - rnd.randrange(4)] ^= 1 << rnd.randrange(8)      # WrongChecksum
            pool.append((bytes(b), None, False, "trailer damaged: " + g[3], 15))
        else:
            b = bytearray(c)
            b[0] ^= 0x07                                                   # BadZlibHeader
            pool.append((bytes(b), None, False, "header damaged: " + g[3], 1))
    return pool, nref


def build_mix(n, raw, uf_comp, uf_off, uf_len, dev):
    """n streams for the mix line: a pool of distinct streams tiled in a fixed shuffled order, packed back
    to back (any alignment is accepted), exact output slots.  -> packed input, offsets, output offsets,
    raw lengths, expectations (python's zlib is the checker here: the inflate output of a valid stream is
    unique)."""
    import random
    import numpy as np
    import torch
    rnd = random.Random(2024)
    pool, nref = mix_pool(raw, uf_comp, uf_off, uf_len, rnd)
    order = [rnd.randrange(len(pool)) for _ in range(n)]
    clen = np.array([len(pool[j][0]) for j in order], dtype=np.int64)
    rlen = np.array([len(pool[j][1]) if pool[j][1] is not None else 65536 for j in order], dtype=np.int64)
    off = np.zeros(n + 1, dtype=np.int64)
    off[1:] = np.cumsum(clen)
    buf = np.zeros(int(off[-1]) + 16, dtype=np.uint8)
    for i, j in enumerate(order):
        buf[off[i]:off[i] + clen[i]] = np.frombuffer(pool[j][0], dtype=np.uint8)
    ooff = np.zeros(n + 1, dtype=np.int64)
    ooff[1:] = np.cumsum(rlen)
    check = [i for i in range(0, n, max(1, n // 97)) if pool[order[i]][2]]
    exp = {"ok": torch.tensor([pool[j][2] for j in order], dtype=torch.bool, device=dev),
           "status": torch.tensor([pool[j][4] for j in order], dtype=torch.int32, device=dev),
           "raw": {i: pool[order[i]][1] for i in check}, "check": check,
           "what": "%d distinct: %d reference corpus / .zz vectors, 48 zlib streams of the bench's buffers (stored, fixed, RLE, "
                   "Huffman-only, levels 6 and 9, short ones), 32 ultra-fast streams, 24 damaged or truncated copies"
                   % (len(pool), nref)}
    return (torch.from_numpy(buf).to(dev), torch.from_numpy(off).to(dev), torch.from_numpy(ooff).to(dev),
            torch.from_numpy(rlen).to(dev), exp)


# ------------------------------------------------------------------------------------------
# CPU baseline (the oracle as the thing timed: allowed here and only here)
# ------------------------------------------------------------------------------------------

def build_native_oracle():
    """A -O3 -march=native build of the oracle for THIS host (the shipped .so is built without
    -march because it travels between machines).  Called BEFORE the process touches the GPU (it
    starts gcc as a child process).  Returns the path or None."""
    import tempfile
    src = os.path.join(ROOT, "oracle", "fdeflate_oracle.c")
    try:
        d = tempfile.mkdtemp(prefix="fdo_native_")
        so = os.path.join(d, "libfdeflate_oracle_native.so")
        subprocess.check_call(["gcc", "-O3", "-march=native", "-fPIC", "-std=gnu11", "-pthread", "-shared",
                               "-o", so, src, "-lz"], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        return so
    except Exception:
        return None


def native_oracle(native_so):
    import ctypes as C
    import oracle_binding as ob
    L, how = None, "gcc -O3"
    if native_so:
        try:
            L, how = C.CDLL(native_so), "gcc -O3 -march=native"
        except OSError:
            L = None
    if L is None:
        L = ob.lib()
    L.fdo_timed_inflate.restype = C.c_double
    L.fdo_timed_inflate.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_int, C.c_int,
                                    C.c_int]
    return L, how


def cpu_general_encode(native_so, host_raw, rle, seconds):
    """Input GB/s of the oracle's level-1 / RLE encoder on ONE host thread over the given buffers
    (numpy uint8 [k, L]), repeated for about `seconds`."""
    import ctypes as C
    import numpy as np
    lib, how = native_oracle(native_so)
    fn = lib.fdo_compress_rle if rle else lib.fdo_compress_level1
    fn.restype = C.c_size_t
    fn.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t]
    k, L = host_raw.shape
    cap = L + L // 2 + 1024
    out = np.empty(cap, dtype=np.uint8)
    done, t0 = 0, time.perf_counter()
    while True:
        for i in range(k):
            if fn(host_raw[i].ctypes.data_as(C.c_void_p), L, out.ctypes.data_as(C.c_void_p), cap) == 0:
                raise RuntimeError("oracle encoder failed")
        done += k
        dt = time.perf_counter() - t0
        if dt >= seconds:
            return done * L / dt / 1e9, how


def effective_cores():
    """Host cores this process may actually use: the affinity mask, capped by the cgroup CPU quota
    (the GPU boxes show 256 logical CPUs but run the job under a 16-CPU quota)."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    quota = None
    try:
        q, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            quota = max(1, int(float(q) / float(period) + 0.5))
    except Exception:
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            period = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                quota = max(1, int(q / period + 0.5))
        except Exception:
            quota = None
    return (min(n, quota) if quota else n), n, quota


def cpu_baseline(args, comp, c_off, native_so):
    """The oracle (a port of the reference algorithm) and system zlib timed on the host cores over
    a bounded sample of the same workload: threads are created once, every thread decodes >= 64
    streams per pass.  Reported next to the GPU number, never the target."""
    import ctypes as C
    import numpy as np
    cores, visible, quota = effective_cores()
    n, L = args.streams, args.stream_bytes
    k = min(n, max(4096, 64 * cores))                     # bounded sample of the same workload
    end = int(c_off[k])
    h_in = np.ascontiguousarray(comp[:end].cpu().numpy())
    h_off = np.ascontiguousarray(c_off[:k + 1].cpu().numpy().astype(np.uint64))
    out = np.zeros(k * L, dtype=np.uint8)                 # touched up front: no page faults inside the timing
    o_off = np.arange(k + 1, dtype=np.uint64) * np.uint64(L)
    lib, how = native_oracle(native_so)

    def timed(threads, kind, k_use):
        def call(passes):
            return lib.fdo_timed_inflate(h_in.ctypes.data_as(C.c_void_p), h_off.ctypes.data_as(C.c_void_p),
                                         out.ctypes.data_as(C.c_void_p), o_off.ctypes.data_as(C.c_void_p),
                                         k_use, threads, passes, kind)
        t1 = call(1)                                       # warm-up + calibration pass
        if t1 <= 0:
            raise RuntimeError("fdo_timed_inflate failed (%r)" % t1)
        passes = max(1, min(1000, int(args.cpu_seconds / t1)))
        t = call(passes)
        if t <= 0:
            raise RuntimeError("fdo_timed_inflate failed (%r)" % t)
        return passes * k_use * L / t / 1e9, passes, t

    k1 = min(k, 1024)                                      # one thread: a smaller slice of the sample
    all_gbs, all_p, all_t = timed(cores, 0, k)
    one_gbs, one_p, one_t = timed(1, 0, k1)
    z_all, _, _ = timed(cores, 1, k)
    z_one, _, _ = timed(1, 1, k1)
    return {"value": round(all_gbs, 3), "unit": "GB/s", "cores": cores, "kind": "port",
            "cpus_visible": visible, "cgroup_cpu_quota": quota,
            "value_1_thread": round(one_gbs, 4),
            "zlib_value": round(z_all, 3), "zlib_value_1_thread": round(z_one, 4),
            "sample": "first %d of the %d streams, %d passes (%.1f GiB decompressed) in %.1f s on %d persistent "
                      "threads (>= %d streams per thread per pass); 1 thread: first %d streams x %d passes in %.1f s; "
                      "oracle/fdo_timed_inflate = C port of the reference algorithm (%s); zlib_* = system zlib "
                      "uncompress() on the same sample"
                      % (k, n, all_p, all_p * k * L / 2**30, all_t, cores, k // cores, k1, one_p, one_t, how)}


# ------------------------------------------------------------------------------------------
# timing helpers
# ------------------------------------------------------------------------------------------

def kernel_source_sha():
    """Hash of the kernel sources -- the .hip files and the headers they include; comment-only and blank lines
    aside: they do not change the machine code --: the PMC traffic figure of profiles/ is only quoted when it was
    measured on exactly this code."""
    h = hashlib.sha256()
    d = os.path.join(ROOT, "fdeflate_amd", "csrc")
    for name in sorted(os.listdir(d)):
        if name.endswith((".h", ".hip", ".inc")):   # (the .cpp files are host code: no kernel in them)
            h.update(name.encode())
            for line in open(os.path.join(d, name), "rb").read().split(b"\n"):
                t = line.strip()
                if t and not t.startswith(b"//"):
                    h.update(t + b"\n")
    return h.hexdigest()[:16]


def profiled_traffic(key):
    tpath = os.path.join(ROOT, "profiles", "traffic_latest.json")
    try:
        tj = json.load(open(tpath))
        if tj.get("kernel_source_sha") != kernel_source_sha():
            return None      # measured on other code: not this run's traffic
        return tj.get(key, {}).get("hbm_bytes_per_step")
    except Exception:
        return None


def time_steps(step, steps, warmup, barrier):
    """warmup untimed steps, then `steps` timed ones -> (wall seconds, per-step HIP-event ms)."""
    import torch
    for _ in range(warmup):
        step()
    barrier()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(steps + 1)]
    barrier()
    t0 = time.perf_counter()
    ev[0].record()
    for k in range(steps):
        step()
        ev[k + 1].record()
    barrier()
    wall = time.perf_counter() - t0
    return wall, [ev[k].elapsed_time(ev[k + 1]) for k in range(steps)]


def roofline(alg_bytes, kern_ms, kernel, traffic):
    achieved = alg_bytes / (kern_ms * 1e-3) / 1e9
    return {"bound": "hbm", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": traffic, "kernel": kernel,
            "kernel_ms_avg": round(kern_ms, 4), "algorithmic_bytes_per_launch": alg_bytes}


DECODE_KERNELS = ("inflate_seg3_kernel (+ inflate_seg2_kernel / inflate_segments_kernel / inflate_canon_kernel / inflate_general_fast_kernel / "
                  "inflate_general_kernel follow-ups on what it leaves over, one fdh_inflate_batch call)")


def main():
    args = parse()
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(launch_ranks(args.gpus))

    import torch
    import torch.distributed as dist
    import fdeflate_amd as fd
    from fdeflate_amd import distributed as fdist
    from fdeflate_amd import synth

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        sys.exit("bench.py: WORLD_SIZE=%d but --gpus %d" % (world, args.gpus))
    want_cpu = rank == 0 and world == 1 and not args.no_cpu_baseline and args.mode == "decode" and args.format == "ultrafast"
    native_so = build_native_oracle() if want_cpu else None   # child process: before any GPU call
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    use_dist = world > 1 or args.force_dist
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        dist.init_process_group("nccl", device_id=dev, rank=rank, world_size=world)

    n, L = args.streams, args.stream_bytes
    raw = synth.gen_batch_torch(rank * n, n, L, device=dev)
    r_off = torch.arange(n + 1, dtype=torch.int64, device=dev) * L
    if args.format == "ultrafast":
        comp, c_off, clen = encode_ultrafast(raw, r_off, dev)
    else:
        comp, c_off, clen = encode_zlib6(raw.cpu().numpy(), dev)
    torch.cuda.synchronize()
    out = torch.empty(n * L, dtype=torch.uint8, device=dev)
    out_len = torch.empty(n, dtype=torch.int32, device=dev)
    status = torch.empty(n, dtype=torch.int32, device=dev)
    adler = torch.empty(n, dtype=torch.int32, device=dev)
    bound = (fd.ultrafast_bound(L) + 15) & ~15
    enc_off = torch.arange(n + 1, dtype=torch.int64, device=dev) * bound
    enc_out = None

    def decode_step():
        fd.inflate_batch(comp, c_off, out, r_off, out_len, status, adler, flags=args.flags)
        if use_dist:
            return fdist.gather_metadata(status, out_len, adler)
        return None

    def encode_step():
        fd.deflate_ultrafast_batch(raw.view(-1), r_off, enc_out, enc_off, out_len)
        if use_dist:
            return fdist.gather_metadata(status, out_len, adler)
        return None

    def barrier():
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    if args.mode == "encode":
        enc_out = torch.empty(n * bound, dtype=torch.uint8, device=dev)
        step = encode_step
    else:
        step = decode_step
        # Correctness outside the timed region: the decode is run and compared with the raw bytes
        # several times over.  (It also puts the GPU under load: an MI355X coming out of idle needs
        # ~30 ms of work before its clocks are steady -- tools/launchtimes.py shows the first eight
        # calls of a row getting faster, 3.64 -> 3.31 ms -- and the W warm-up steps asked for may be
        # shorter than that.)
        for _ in range(8):
            out.zero_()
            decode_step()
            assert int(status.abs().sum()) == 0, "decode reported errors"
            assert bool((out_len == L).all()) and torch.equal(out, raw.view(-1)), "decoded bytes differ"
        barrier()

    wall, kern_ms = time_steps(step, args.steps, args.warmup, barrier)
    t = torch.tensor([wall], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    wall = float(t[0])

    payload_ms = None
    if world > 1 and args.mode == "decode" and not args.no_payload_gather:
        # the optional payload gather (every rank receives every shard's decoded bytes): link-bound
        # over xGMI, reported on its own and never part of `value`
        try:
            full = fdist.gather_payload(out)   # warm-up (allocates world x shard)
            barrier()
            t0 = time.perf_counter()
            fdist.gather_payload(out, into=full)
            barrier()
            tp = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=dev)
            dist.all_reduce(tp, op=dist.ReduceOp.MAX)
            payload_ms = round(float(tp[0]) * 1e3, 3)
            del full
        except Exception as e:
            payload_ms = "failed: %r" % (e,)

    in_bytes = int(clen.sum())
    out_bytes = n * L
    ms_per_step = wall * 1e3 / args.steps
    if args.mode == "decode":
        metric = "decompressed GB/s, batched zlib decode of %d x %d KiB PNG-filter streams per GPU" % (n, L // 1024)
    else:
        metric = "input GB/s, batched ultra-fast zlib encode of %d x %d KiB buffers per GPU" % (n, L // 1024)
    value = out_bytes * world / (wall / args.steps) / 1e9

    if rank == 0:
        kern_avg_ms = sum(kern_ms) / len(kern_ms)
        alg = in_bytes + out_bytes + 24 * n    # in_len + out_len + 24 per stream
        traffic = profiled_traffic(args.mode) if args.format == "ultrafast" and n == 65536 else None
        res = {
            "metric": metric, "value": round(value, 3), "unit": "GB/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u8",
            "data": "synthetic",
            "config": {"workload": "BASELINE config 2: %d independent %d KiB PNG-filter zlib streams per GPU, "
                                   "%s, %s, mean compressed %.0f B/stream"
                                   % (n, L // 1024, "ultra-fast format (fdeflate HEADER, dist-1 runs)"
                                      if args.format == "ultrafast" else "zlib level 6", args.mode,
                                      in_bytes / n),
                       "streams_per_gpu": n, "stream_bytes": L, "format": args.format, "mode": args.mode,
                       "sharding": "contiguous stream ranges per rank, metadata all_gather per step"},
            "roofline": roofline(alg, kern_avg_ms, DECODE_KERNELS if args.mode == "decode" else "deflate_ultrafast_kernel",
                                 traffic),
        }
        res["roofline"]["traffic_source"] = ("profiles/traffic_latest.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE "
                                             "passes of this exact kernel source)" if traffic is not None else None)
        res["roofline"]["kernel_ms_min"] = round(min(kern_ms), 4)
        res["roofline"]["kernel_ms_max"] = round(max(kern_ms), 4)
        if payload_ms is not None:
            res["payload_gather_ms"] = payload_ms
        also = []
        sel = set(x for x in args.also_select.split(",") if x)
        full = n == 65536
        if world == 1 and not args.no_also and args.mode == "decode" and args.format == "ultrafast":
            # BASELINE config 3: ultra-fast encode of the same buffers
            try:
                if "encode" not in sel:
                    raise KeyError("skipped")
                enc_out = torch.empty(n * bound, dtype=torch.uint8, device=dev)
                w2, k2 = time_steps(encode_step, args.steps, args.warmup, barrier)
                elen = out_len.to(torch.int64)
                assert bool((elen == clen).all()), "encoder lengths changed"
                k2avg = sum(k2) / len(k2)
                also.append({"workload": "BASELINE config 3: ultra-fast encode of the same %d x %d KiB buffers" % (n, L // 1024),
                             "metric": "input GB/s", "value": round(out_bytes / (w2 / args.steps) / 1e9, 3),
                             "ms_per_step": round(w2 * 1e3 / args.steps, 4),
                             "roofline": roofline(alg, k2avg, "deflate_ultrafast_kernel", profiled_traffic("encode")
                                                  if n == 65536 else None)})
                del enc_out
            except KeyError:
                pass
            except Exception as e:
                also.append({"workload": "BASELINE config 3 (encode)", "error": repr(e)})
            # the general encoder (level 1, RLE) on the same buffers
            for mode, label in ((fd.MODE_LEVEL1, "compress_to_vec (level 1, greedy parse + hash table)"),
                                (fd.MODE_RLE, "compress_to_vec_rle (RLE parse)")):
                try:
                    key = "level1" if mode == fd.MODE_LEVEL1 else "rle"
                    if key not in sel:
                        continue
                    ng = min(n, args.general_streams)
                    gbound = (fd.compress_bound(L) + 15) & ~15
                    g_off = torch.arange(ng + 1, dtype=torch.int64, device=dev) * gbound
                    gout = torch.empty(ng * gbound, dtype=torch.uint8, device=dev)
                    g_in, gr_off = raw[:ng].view(-1), r_off[:ng + 1]
                    glen = fd.deflate_general_batch(g_in, gr_off, gout, g_off, mode)   # (returns when done)
                    import zlib as _z
                    for i in (0, ng // 2, ng - 1):   # sanity: the streams inflate back to the input
                        c = gout[i * gbound:i * gbound + int(glen[i])].cpu().numpy().tobytes()
                        assert _z.decompress(c) == raw[i].cpu().numpy().tobytes(), "general encoder stream %d" % i
                    gsteps = 3
                    barrier()
                    t0 = time.perf_counter()
                    for _ in range(gsteps):
                        fd.deflate_general_batch(g_in, gr_off, gout, g_off, mode, glen)
                    barrier()
                    gw = time.perf_counter() - t0
                    galg = ng * L + int(glen.to(torch.int64).sum()) + 24 * ng
                    entry = {"workload": "SURVEY 8f: %s of %d x %d KiB buffers" % (label, ng, L // 1024),
                             "metric": "input GB/s", "value": round(ng * L / (gw / gsteps) / 1e9, 3),
                             "ms_per_step": round(gw * 1e3 / gsteps, 4), "steps": gsteps,
                             "ratio": round(float(glen.to(torch.int64).sum()) / (ng * L), 4),
                             # (host clock around a call that returns when the kernels are done)
                             "roofline": roofline(galg, gw * 1e3 / gsteps, "deflate_parse_kernel + deflate_write_kernel",
                                                  profiled_traffic(key) if ng == args.general_streams == 65536 else None)}
                    if want_cpu:
                        v, how = cpu_general_encode(native_so, raw[:64].cpu().numpy(), mode == fd.MODE_RLE, 1.5)
                        entry["cpu_port_1_thread"] = {"value": round(v, 4), "unit": "GB/s", "kind": "port",
                                                      "sample": "the first 64 buffers, repeated for 1.5 s; oracle (%s)" % how}
                    also.append(entry)
                    del gout
                except Exception as e:
                    also.append({"workload": "SURVEY 8f: %s" % label, "error": repr(e)})
            # decode + PNG scanline reconstruction in one call (the buffers are
            # 64 scanlines of 1023 bytes behind a filter-type byte each, RGB8)
            try:
                if "png" not in sel:
                    raise KeyError("skipped")
                from fdeflate_amd import synth as _synth
                rb_png, bpp_png = _synth.ROW_BYTES - 1, 3
                rows_png = L // _synth.ROW_BYTES
                assert rows_png * _synth.ROW_BYTES == L
                p_off = torch.arange(n + 1, dtype=torch.int64, device=dev) * (rows_png * rb_png)
                pix = torch.empty(n * rows_png * rb_png, dtype=torch.uint8, device=dev)

                def png_step():
                    return fd.inflate_png_batch(comp, c_off, out, r_off, pix, p_off, rb_png, bpp_png)

                _, pst_status, _, pst = png_step()
                barrier()
                assert int(pst_status.abs().sum()) == 0 and int(pst.abs().sum()) == 0, "png decode statuses"
                # property check: filtering the reconstructed pixels with the rows' own filter types
                # gives the decoded scanlines back
                types = out.view(n, rows_png, rb_png + 1)[:, :, 0].contiguous().view(-1)
                t_off = torch.arange(n + 1, dtype=torch.int64, device=dev) * rows_png
                back = torch.empty_like(out)
                fst = fd.png_filter_batch(pix, p_off, types, t_off, back, r_off, rb_png, bpp_png)
                assert int(fst.abs().sum()) == 0 and torch.equal(back, out), "filter(reconstruct(x)) != x"
                del back, types
                psteps = max(3, args.steps // 4)
                w4, k4 = time_steps(png_step, psteps, 1, barrier)
                # algorithmic bytes: the decode's + the reconstruction's (scanlines read once, pixels written once)
                palg = alg + n * L + n * rows_png * rb_png
                also.append({"workload": "SURVEY 8f: inflate + PNG scanline reconstruction (fdh_inflate_png_batch) of the same "
                                         "%d streams, %d rows x %d bytes, %d bytes per pixel" % (n, rows_png, rb_png, bpp_png),
                             "metric": "decompressed GB/s", "value": round(n * L / (w4 / psteps) / 1e9, 3),
                             "ms_per_step": round(w4 * 1e3 / psteps, 4), "steps": psteps,
                             "roofline": roofline(palg, sum(k4) / len(k4), "inflate_seg3_kernel + png_pipe_kernel",
                                                  profiled_traffic("png") if full else None)})
                # ... and the other direction: filtering fused into the ultra-fast encoder (the pixels just
                # reconstructed, the rows' own filter types): must give the bench's compressed streams back
                try:
                    types = out.view(n, rows_png, rb_png + 1)[:, :, 0].contiguous().view(-1)
                    t_off = torch.arange(n + 1, dtype=torch.int64, device=dev) * rows_png
                    fenc = torch.empty(n * bound, dtype=torch.uint8, device=dev)

                    def fenc_step():
                        return fd.png_filter_deflate_ultrafast_batch(pix, p_off, types, t_off, fenc, enc_off, rb_png, bpp_png)

                    flen, fstat = fenc_step()
                    barrier()
                    assert int(fstat.abs().sum()) == 0 and torch.equal(flen.to(torch.int64), clen), "fused filter + encode lengths"
                    for i in (0, 7, 15, n - 1):
                        a0, b0 = i * bound, int(c_off[i])
                        assert torch.equal(fenc[a0:a0 + int(clen[i])], comp[b0:b0 + int(clen[i])]), "fused filter + encode bytes"
                    w5, k5 = time_steps(fenc_step, psteps, 1, barrier)
                    falg = n * rows_png * rb_png + n * rows_png + in_bytes + 8 * n
                    also.append({"workload": "SURVEY 8f: PNG filtering fused into the ultra-fast encoder "
                                             "(fdh_png_filter_deflate_ultrafast_batch) of the %d images" % n,
                                 "metric": "input GB/s", "value": round(n * rows_png * rb_png / (w5 / psteps) / 1e9, 3),
                                 "ms_per_step": round(w5 * 1e3 / psteps, 4), "steps": psteps,
                                 "roofline": roofline(falg, sum(k5) / len(k5), "deflate_ultrafast_kernel_t<true>",
                                                      profiled_traffic("filterenc") if full else None)})
                    del fenc, types
                except Exception as e:
                    also.append({"workload": "SURVEY 8f: PNG filter + ultra-fast encode", "error": repr(e)})
                del pix
            except KeyError:
                pass
            except Exception as e:
                also.append({"workload": "SURVEY 8f: inflate + PNG reconstruction", "error": repr(e)})
            # BASELINE config 2 (ii): the same data as zlib level-6 streams (general kernels)
            try:
                if "zlib6" not in sel:
                    raise KeyError("skipped")
                nz = min(n, args.zlib6_streams)
                zcomp, zoff, zlen = encode_zlib6(raw[:nz].cpu().numpy(), dev)
                zr_off = r_off[:nz + 1]
                zout = out[:nz * L]

                def z_step():
                    fd.inflate_batch(zcomp, zoff, zout, zr_off, out_len[:nz], status[:nz], adler[:nz])

                z_step()
                barrier()
                assert int(status[:nz].abs().sum()) == 0 and torch.equal(zout, raw[:nz].view(-1)), "zlib-6 decode differs"
                zsteps = max(3, args.steps // 4)
                w3, k3 = time_steps(z_step, zsteps, 1, barrier)
                zalg = int(zlen.sum()) + nz * L + 24 * nz
                also.append({"workload": "BASELINE config 2 (ii): %d x %d KiB of the same data as zlib level-6 streams "
                                         "(dynamic blocks, real distances), mean compressed %.0f B/stream"
                                         % (nz, L // 1024, float(zlen.sum()) / nz),
                             "metric": "decompressed GB/s", "value": round(nz * L / (w3 / zsteps) / 1e9, 3),
                             "ms_per_step": round(w3 * 1e3 / zsteps, 4), "steps": zsteps,
                             "roofline": roofline(zalg, sum(k3) / len(k3),
                                                  "inflate_lz_kernel (the kernels in front pass the streams on; nothing is left for the tile decoders behind)",
                                                  profiled_traffic("zlib6") if nz == 65536 else None)})
            except KeyError:
                pass
            except Exception as e:
                also.append({"workload": "BASELINE config 2 (ii) (zlib-6 decode)", "error": repr(e)})
            # BASELINE config 5 / SURVEY 8(d) C2 (iii): the mix -- the reference's corpus and .zz vectors, stored /
            # fixed / dynamic / RLE / Huffman-only streams of the bench's buffers, ultra-fast streams, damaged and
            # truncated copies -- tiled to n streams in one batch; Ok streams checked against the raw bytes
            try:
                if "mix" not in sel:
                    raise KeyError("skipped")
                mcomp, moff, mr_off, mraw_len, mexp = build_mix(n, raw, comp, c_off, clen, dev)
                mout = out[:int(mr_off[-1])]
                mst, mln, mad = status[:n], out_len[:n], adler[:n]

                def m_step():
                    fd.inflate_batch(mcomp, moff, mout, mr_off, mln, mst, mad)

                m_step()
                barrier()
                ok = mst == 0
                assert bool((ok == mexp["ok"]).all()), "mix: a stream's Ok / not-Ok differs from what its maker expects"
                assert bool((mst == mexp["status"]).all()), "mix: the error kind of a damaged stream differs from what it was built for"
                assert bool((mln[ok].to(torch.int64) == mraw_len[ok]).all()), "mix: length of an Ok stream"
                for i in mexp["check"]:
                    a0 = int(mr_off[i])
                    assert mout[a0:a0 + int(mraw_len[i])].cpu().numpy().tobytes() == mexp["raw"][i], "mix: bytes of stream %d" % i
                msteps = max(3, args.steps // 4)
                w6, k6 = time_steps(m_step, msteps, 1, barrier)
                mbytes = int(mln[ok].to(torch.int64).sum())
                malg = int((moff[1:] - moff[:-1]).sum()) + mbytes + 24 * n
                also.append({"workload": "BASELINE config 5 / SURVEY 8(d) C2 (iii): mix of %d streams in one batch (%s)" % (n, mexp["what"]),
                             "metric": "decompressed GB/s", "value": round(mbytes / (w6 / msteps) / 1e9, 3),
                             "ms_per_step": round(w6 * 1e3 / msteps, 4), "steps": msteps,
                             "ok_streams": int(ok.sum()), "other_streams": int((~ok).sum()),
                             "roofline": roofline(malg, sum(k6) / len(k6), "one fdh_inflate_batch call: every kernel of the pipeline",
                                                  profiled_traffic("mix") if full else None)})
                del mcomp, mout
            except KeyError:
                pass
            except Exception as e:
                also.append({"workload": "BASELINE config 5 (mix)", "error": repr(e)})
        if also:
            res["also"] = also
        if want_cpu:
            try:
                res["cpu_baseline"] = cpu_baseline(args, comp, c_off, native_so)
            except Exception as e:  # the baseline is a reported extra, never fatal
                res["cpu_baseline"] = {"value": None, "unit": "GB/s", "cores": os.cpu_count(), "kind": "port",
                                       "sample": "failed: %r" % (e,)}
        # RCCL prints a version banner through C stdio, which is flushed at exit -- i.e. behind a line
        # printed from Python.  Flush it out first so that the JSON line is the last line of stdout.
        try:
            import ctypes
            ctypes.CDLL(None).fflush(None)
        except Exception:
            pass
        sys.stdout.flush()
        print(json.dumps(res), flush=True)
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
