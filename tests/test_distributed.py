"""The N > 1 path on CPU: world_size-2 gloo processes shard a batch by contiguous stream ranges,
decode their shard (the oracle stands in for the GPU kernel here) and all-gather the per-stream
metadata exactly as bench.py does over RCCL.  No data-path collective exists to test."""
import os
import socket
import sys
import zlib

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _make_batch(n):
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    sys.path.insert(0, ROOT)
    import oracle_binding as ob
    from fdeflate_amd import synth
    raws = [synth.gen_stream_np(i, 2048).tobytes() for i in range(n)]
    comps = [ob.compress_ultra_fast(r) for r in raws]
    comps[3] = comps[3][:-7]           # truncated -> InsufficientInput
    bad = bytearray(comps[5])
    bad[-1] ^= 1
    comps[5] = bytes(bad)              # WrongChecksum
    return raws, comps


def _worker(rank, world, port, n, q):
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import oracle_binding as ob
    from fdeflate_amd import distributed as fdist
    raws, comps = _make_batch(n)
    lo, hi = fdist.shard_range(n, rank, world)
    per = (n + world - 1) // world
    status = torch.zeros(per, dtype=torch.int32)
    out_len = torch.zeros(per, dtype=torch.int32)
    adler = torch.zeros(per, dtype=torch.int32)
    for k, i in enumerate(range(lo, hi)):
        st, out, ad = ob.decompress_bounded(comps[i], 2048)
        status[k], out_len[k] = st, len(out)
        adler[k] = np.int32(np.uint32(ad).view(np.int32))
    meta = fdist.gather_metadata(status, out_len, adler)
    if rank == 0:
        q.put(meta.numpy().copy())
    dist.barrier()
    dist.destroy_process_group()


def test_shard_range_partitions_every_stream_once():
    sys.path.insert(0, ROOT)
    from fdeflate_amd import distributed as fdist
    for n in (0, 1, 7, 8, 65536, 1048576 + 3):
        for world in (1, 2, 4, 8):
            ranges = [fdist.shard_range(n, r, world) for r in range(world)]
            assert ranges[0][0] == 0 and ranges[-1][1] == n
            for (a0, a1), (b0, b1) in zip(ranges, ranges[1:]):
                assert a1 == b0 and a0 <= a1
            assert max(hi - lo for lo, hi in ranges) <= (n + world - 1) // world


def test_two_rank_gloo_shard_decode_gather():
    world, n = 2, 16
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n, q)) for r in range(world)]
    for p in procs:
        p.start()
    meta = q.get(timeout=120)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    # single-process truth
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_binding as ob
    raws, comps = _make_batch(n)
    per = n // world
    assert meta.shape == (world, 3, per)
    for i in range(n):
        st, out, ad = ob.decompress_bounded(comps[i], 2048)
        r, k = divmod(i, per)
        assert int(meta[r, 0, k]) == st
        assert int(meta[r, 1, k]) == len(out)
        if st == 0:
            assert int(np.uint32(np.int32(meta[r, 2, k]))) == zlib.adler32(raws[i])
    assert int(meta[0, 0, 3]) == 2 and int(meta[0, 0, 5]) == 15


def test_bench_launcher_starts_its_own_ranks_and_reports_failure():
    """`python bench.py --gpus 2` with no rank environment starts two rank processes itself (before
    any GPU call, free port) and ends when a rank fails: without a GPU in this container both ranks
    fail at once, the parent must return non-zero promptly instead of waiting at a collective."""
    import subprocess
    import time
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    t0 = time.time()
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0",
                        "--streams", "64", "--no-cpu-baseline", "--no-also"], env=env, capture_output=True, timeout=300)
    if torch.cuda.is_available() and torch.cuda.device_count() >= 2:
        assert p.returncode == 0 and b'"n_gpus": 2' in p.stdout
    else:
        assert p.returncode != 0
        assert time.time() - t0 < 280


@pytest.mark.gpu
def test_one_rank_nccl_group_gathers_on_the_gpu():
    """The branch `bench.py --gpus N` takes (torch.distributed backend "nccl" = RCCL, all_gather_into_tensor) with
    one rank, in this process: shard, decode on the GPU through the C ABI, gather the per-stream results and a
    payload with fdist.gather_metadata / gather_payload, compare with the oracle."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import fdeflate_amd as fd
    import oracle_binding as ob
    from fdeflate_amd import distributed as fdist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(_free_port())
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    dist.init_process_group("nccl", device_id=dev, rank=0, world_size=1)
    try:
        assert dist.get_backend() == "nccl"
        n = 16
        raws, comps = _make_batch(n)
        lo, hi = fdist.shard_range(n, 0, 1)
        assert (lo, hi) == (0, n)
        blob = b"".join(comps)
        in_off = np.zeros(n + 1, dtype=np.int64)
        in_off[1:] = np.cumsum([len(c) for c in comps])
        out_off = np.arange(n + 1, dtype=np.int64) * 2048
        d_in = torch.from_numpy(np.frombuffer(blob, dtype=np.uint8).copy()).to(dev)
        d_out = torch.zeros(n * 2048, dtype=torch.uint8, device=dev)
        out_len, status, adler = fd.inflate_batch(d_in, torch.from_numpy(in_off).to(dev), d_out, torch.from_numpy(out_off).to(dev))
        meta = fdist.gather_metadata(status, out_len, adler)
        full = fdist.gather_payload(d_out)
        torch.cuda.synchronize()
        assert meta.is_cuda and tuple(meta.shape) == (1, 3, n) and tuple(full.shape) == (1, n * 2048)
        meta = meta.cpu().numpy()
        h = full.cpu().numpy()[0]
        for i in range(n):
            st, out, ad = ob.decompress_bounded(comps[i], 2048)
            assert int(meta[0, 0, i]) == st
            if st in (0, 17):
                assert int(meta[0, 1, i]) == len(out)
            if st == 0:
                assert int(np.uint32(np.int32(meta[0, 2, i]))) == ad
                assert h[2048 * i:2048 * i + len(out)].tobytes() == out
    finally:
        dist.destroy_process_group()
