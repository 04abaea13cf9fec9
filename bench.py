#!/usr/bin/env python3
"""bench.py -- decompressed GB/s of the batched zlib decode (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W

One "step" = one pass of fdh_inflate_batch over the whole per-GPU batch (65 536 independent
64 KiB PNG-filter zlib streams in the ultra-fast format, BASELINE config 2), inputs resident in
HBM.  For N > 1 the driver launches one rank per GPU (torch.distributed / RCCL); streams are
sharded by rank with no data-path collective, the per-stream metadata is all-gathered inside
the step, and the step time is the max over ranks.  Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
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
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--force-dist", action="store_true",
                    help="testing: run the RCCL metadata gather even with one rank")
    return ap.parse_args()


def build_inputs(args, rank, dev):
    """raw [n, L] on the device and its ultra-fast encoding packed 16-B aligned."""
    import torch
    import fdeflate_amd as fd
    from fdeflate_amd import synth
    n, L = args.streams, args.stream_bytes
    raw = synth.gen_batch_torch(rank * n, n, L, device=dev)
    r_off = torch.arange(n + 1, dtype=torch.int64, device=dev) * L
    if args.format == "ultrafast":
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
    else:
        import zlib
        import numpy as np
        h = raw.cpu().numpy()
        blobs = [zlib.compress(h[i].tobytes(), 6) for i in range(n)]
        clen_h = np.array([len(b) for b in blobs], dtype=np.int64)
        off_h = np.zeros(n + 1, dtype=np.int64)
        off_h[1:] = np.cumsum((clen_h + 15) & ~15)
        buf = np.zeros(int(off_h[-1]), dtype=np.uint8)
        for i, b in enumerate(blobs):
            buf[off_h[i]:off_h[i] + len(b)] = np.frombuffer(b, dtype=np.uint8)
        comp = torch.from_numpy(buf).to(dev)
        c_off = torch.from_numpy(off_h).to(dev)
        clen = torch.from_numpy(clen_h).to(dev)
    torch.cuda.synchronize()
    return raw, r_off, comp, c_off, clen


def cpu_baseline(args, raw, comp, c_off, clen):
    """The oracle (a port of the reference algorithm) timed on the host cores over a bounded
    sample of the same workload.  Reported next to the GPU number, never the target."""
    import numpy as np
    import oracle_binding as ob
    cores = os.cpu_count() or 1
    n = args.streams
    L = args.stream_bytes

    k = min(n, 4096)                     # bounded sample of the same workload
    end = int(c_off[k])
    h_in = comp[:end].cpu().numpy()
    h_off = c_off[:k + 1].cpu().numpy().astype(np.uint64)
    out = np.zeros(k * L, dtype=np.uint8)    # touched up front: no page faults inside the timing
    o_off = (np.arange(k + 1, dtype=np.uint64) * np.uint64(L))
    ob.inflate_batch(h_in, h_off, out, o_off, False, cores)   # warm-up pass (threads, caches)
    passes, t_total = 0, 0.0
    while t_total < args.cpu_seconds and passes < 200:
        t0 = time.perf_counter()
        out_len, status, adler = ob.inflate_batch(h_in, h_off, out, o_off, False, cores)
        t_total += time.perf_counter() - t0
        passes += 1
        assert int(status.sum()) == 0 and int(out_len.sum()) == k * L
    gbs = passes * k * L / t_total / 1e9
    k2, dt2 = k * passes, t_total
    return {"value": round(gbs, 3), "unit": "GB/s", "cores": cores, "kind": "port",
            "sample": "first %d of the %d streams decoded %d times (%.1f GiB) in %.1f s, "
                      "oracle/fdo_inflate_batch (C restatement of the reference), %d threads"
                      % (k, n, passes, k2 * L / 2**30, dt2, cores)}


def main():
    args = parse()
    import torch
    import torch.distributed as dist
    import fdeflate_amd as fd
    from fdeflate_amd import distributed as fdist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    assert world == args.gpus, "launch with torch.distributed.run --nproc-per-node %d" % args.gpus
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    use_dist = world > 1 or args.force_dist
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        dist.init_process_group("nccl", device_id=dev, rank=rank, world_size=world)

    n, L = args.streams, args.stream_bytes
    raw, r_off, comp, c_off, clen = build_inputs(args, rank, dev)
    out = torch.empty(n * L, dtype=torch.uint8, device=dev)
    out_len = torch.empty(n, dtype=torch.int32, device=dev)
    status = torch.empty(n, dtype=torch.int32, device=dev)
    adler = torch.empty(n, dtype=torch.int32, device=dev)
    bound = (fd.ultrafast_bound(L) + 15) & ~15
    if args.mode == "encode":
        enc_off = torch.arange(n + 1, dtype=torch.int64, device=dev) * bound
        enc_out = torch.empty(n * bound, dtype=torch.uint8, device=dev)

    def step():
        if args.mode == "decode":
            fd.inflate_batch(comp, c_off, out, r_off, out_len, status, adler, flags=args.flags)
        else:
            fd.deflate_ultrafast_batch(raw.view(-1), r_off, enc_out, enc_off, out_len)
        if use_dist:
            return fdist.gather_metadata(status, out_len, adler)
        return None

    def barrier():
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    barrier()
    # correctness outside the timed region
    if args.mode == "decode":
        assert int(status.abs().sum()) == 0, "decode reported errors"
        assert bool((out_len == L).all()) and torch.equal(out, raw.view(-1)), "decoded bytes differ"
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]
    barrier()
    t0 = time.perf_counter()
    ev[0].record()
    for k in range(args.steps):
        step()
        ev[k + 1].record()
    barrier()
    wall = time.perf_counter() - t0
    kern_ms = [ev[k].elapsed_time(ev[k + 1]) for k in range(args.steps)]
    t = torch.tensor([wall], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    wall = float(t[0])

    in_bytes = int(clen.sum())
    out_bytes = n * L
    ms_per_step = wall * 1e3 / args.steps
    if args.mode == "decode":
        units = out_bytes          # decompressed bytes
        metric = "decompressed GB/s, batched zlib decode of %d x %d KiB PNG-filter streams per GPU" % (n, L // 1024)
    else:
        units = out_bytes          # raw input bytes consumed
        metric = "input GB/s, batched ultra-fast zlib encode of %d x %d KiB buffers per GPU" % (n, L // 1024)
    value = units * world / (wall / args.steps) / 1e9

    if rank == 0:
        kern_avg_ms = sum(kern_ms) / len(kern_ms)
        alg = in_bytes + out_bytes + 24 * n    # in_len + out_len + 24 per stream
        achieved = alg / (kern_avg_ms * 1e-3) / 1e9
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "traffic_latest.json")
        if os.path.exists(tpath):
            try:
                tj = json.load(open(tpath))
                # the committed PMC passes were taken on the default (ultra-fast format) workload only
                traffic = tj.get(args.mode, {}).get("hbm_bytes_per_step") if args.format == "ultrafast" else None
            except Exception:
                traffic = None
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
            "roofline": {"bound": "hbm", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": traffic,
                         "kernel": "inflate_segments_kernel (+ inflate_canon_kernel / inflate_general_fast_kernel / inflate_general_kernel follow-ups, one fdh_inflate_batch launch)"
                         if args.mode == "decode" else "deflate_ultrafast_kernel",
                         "kernel_ms_avg": round(kern_avg_ms, 4), "algorithmic_bytes_per_launch": alg},
        }
        if world == 1 and not args.no_cpu_baseline and args.mode == "decode":
            try:
                res["cpu_baseline"] = cpu_baseline(args, raw, comp, c_off, clen)
            except Exception as e:  # the baseline is a reported extra, never fatal
                res["cpu_baseline"] = {"value": None, "unit": "GB/s", "cores": os.cpu_count(), "kind": "port",
                                       "sample": "failed: %r" % (e,)}
        print(json.dumps(res))
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
